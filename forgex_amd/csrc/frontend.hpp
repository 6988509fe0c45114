// Host-side front end of the Forgex match path: pattern text -> syntax tree -> literals -> range NFA.
//
// This is the "host compile feeding the table" part of the hot path.
// It follows the reference's behaviour statement by statement, quirks included, because the
// accepted language (and therefore every match result) is defined by it:
//   tokenizer        reference src/ast/syntax_tree_node_m.F90:133-215
//   parser           reference src/ast/syntax_tree_graph_m.F90:61-95, :205-1213
//   class parser     reference src/ast/character_array_m.F90:45-332
//   literal factors  reference src/ast/syntax_tree_optimize_m.F90:42-360
//   NFA build        reference src/nfa/nfa_node_m.F90:61-572, src/essential/segment_disjoin_m.F90:36-182
//   UTF-8 / segments reference src/essential/utf8_m.f90:44-438, src/essential/segment_m.F90:37-506
// No reference code is copied; indices are kept 1-based where that keeps the arithmetic legible.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace fxfe {

// ---- constants (reference src/essential/parameters_m.f90:15-154) -------------------------------
constexpr int INVALID_INDEX = -9999;
constexpr int INVALID_CHAR_INDEX = -9999;
constexpr int INVALID_REPEAT_VAL = -9999;
constexpr int INFINITE_REPEAT = -9998;
constexpr int UTF8_CODE_MAX = 1114111;
constexpr int UTF8_CODE_MIN = 32;
constexpr int UTF8_CODE_EMPTY = 0;
constexpr int NFA_NULL_TRANSITION = -1;
constexpr int NFA_C_SIZE = 16;
constexpr int TREE_NODE_HARD_LIMIT = 2048;

// status codes (reference src/essential/error_m.F90:12-38)
enum Status : int {
   SYNTAX_VALID = 0,
   SYNTAX_ERR,
   SYNTAX_ERR_PARENTHESIS_MISSING,
   SYNTAX_ERR_PARENTHESIS_UNEXPECTED,
   SYNTAX_ERR_BRACKET_MISSING,
   SYNTAX_ERR_BRACKET_UNEXPECTED,
   SYNTAX_ERR_CURLYBRACE_MISSING,
   SYNTAX_ERR_CURLYBRACE_UNEXPECTED,
   SYNTAX_ERR_INVALID_TIMES,
   SYNTAX_ERR_ESCAPED_SYMBOL_MISSING,
   SYNTAX_ERR_ESCAPED_SYMBOL_INVALID,
   SYNTAX_ERR_EMPTY_CHARACTER_CLASS,
   SYNTAX_ERR_RANGE_WITH_ESCAPE_SEQUENCES,
   SYNTAX_ERR_MISPLACED_SUBTRACTION_OPERATOR,
   SYNTAX_ERR_INVALID_CHARACTER_RANGE,
   SYNTAX_ERR_CHAR_CLASS_SUBTRANCTION_NOT_IMPLEMENTED,
   SYNTAX_ERR_STAR_INCOMPLETE,
   SYNTAX_ERR_PLUS_INCOMPLETE,
   SYNTAX_ERR_QUESTION_INCOMPLETE,
   SYNTAX_ERR_INVALID_HEXADECIMAL,
   SYNTAX_ERR_HEX_DIGITS_NOT_ENOUGH,
   SYNTAX_ERR_UNICODE_EXCEED,
   SYNTAX_ERR_UNICODE_PROPERTY_NOT_IMPLEMENTED,
   SYNTAX_ERR_THIS_SHOULD_NOT_HAPPEN,
   ALLOCATION_ERR,
   // ---- codes beyond the reference's enum: where the reference executes `error stop`
   //       this library returns a status instead of aborting the process.
   FX_ERR_TREE_LIMIT = 100,   // > 2048 syntax-tree nodes (reference syntax_tree_graph_m.F90:115-117)
   FX_ERR_NFA_LIMIT = 101,    // NFA larger than this build accepts
   FX_ERR_DFA_LIMIT = 102,    // > 16385 DFA states (reference lazy_dfa_graph_m.F90:90-92)
   FX_ERR_UNDEFINED = 103,    // pattern drives the reference into out-of-bounds accesses (undefined there)
};
const char* status_message(int code);   // reference error_m.F90:127-211

// ---- segments ------------------------------------------------------------------------------------
struct Seg {
   int32_t min = UTF8_CODE_MAX + 2;
   int32_t max = UTF8_CODE_MAX + 2;
   Seg() = default;
   Seg(int32_t a, int32_t b) : min(a), max(b) {}
   bool operator==(const Seg& o) const { return min == o.min && max == o.max; }
   bool operator!=(const Seg& o) const { return !(*this == o); }
   bool validate() const;   // segment_m.F90:185-193
};
extern const Seg SEG_INIT, SEG_ERROR, SEG_EPSILON, SEG_EMPTY, SEG_ANY, SEG_TAB, SEG_LF, SEG_FF, SEG_CR,
   SEG_SPACE, SEG_UNDERSCORE, SEG_DIGIT, SEG_UPPERCASE, SEG_LOWERCASE, SEG_ZENKAKU_SPACE, SEG_UPPER, SEG_WHOLE;

void sort_segment_by_min(std::vector<Seg>& s);
void merge_segments(std::vector<Seg>& s);
void invert_segment_list(std::vector<Seg>& s);
void disjoin(std::vector<Seg>& list);

// ---- UTF-8 (byte indexed, 1-based like the reference) ---------------------------------------------
int idxutf8(const std::string& s, int curr);
int next_idxutf8(const std::string& s, int curr);
bool is_valid_multiple_byte_character(const std::string& ch);
int32_t ichar_utf8(const std::string& ch);
std::string char_utf8(int32_t code);
int len_utf8(const std::string& s);
std::string reverse_utf8(const std::string& s);
// text-side strict step: returns next index and validity (utf8_m.f90:168-191)
void next_idxutf8_strict(const std::string& s, int curr, int& next, bool& valid);

// ---- syntax tree ------------------------------------------------------------------------------------
enum Op : int { op_not_init = 0, op_char, op_concat, op_union, op_closure, op_repeat, op_empty };

struct TreeNode {
   int op = op_not_init;
   std::vector<Seg> c;
   bool has_c = false;
   int left_i = INVALID_INDEX, right_i = INVALID_INDEX, parent_i = INVALID_INDEX, own_i = INVALID_INDEX;
   int min_repeat = 0, max_repeat = 0;
};

struct Tree {
   std::vector<TreeNode> nodes;   // 1-based: nodes[0] unused
   int top = 0;
   bool is_valid = true;
   int code = SYNTAX_VALID;
   void build(const std::string& pattern);   // syntax_tree_graph_m.F90:61-95
};

struct Literals {
   std::string all, prefix, suffix;
};
Literals extract_literal(const Tree& t);   // syntax_tree_optimize_m.F90:42-55

// ---- NFA ---------------------------------------------------------------------------------------------
struct NfaTransition {
   std::vector<Seg> c;     // every entry takes part in membership / epsilon tests, as in the reference
   int c_top = 0;
   int dst = NFA_NULL_TRANSITION;
   bool is_epsilon() const;                 // any(c == SEG_EPSILON)  (automaton_m.F90:145, nfa_graph_m.F90:97)
   bool accepts(int32_t code) const;        // symbol .in. segs       (automaton_m.F90:251)
};
struct NfaNode {
   std::vector<NfaTransition> forward;      // registered transitions only
};
struct Nfa {
   std::vector<NfaNode> nodes;   // 1-based
   int nfa_top = 0;
   int entry = 0, exit = 0;
   std::vector<Seg> all_segments;
   int status = SYNTAX_VALID;    // FX_ERR_NFA_LIMIT when the build was abandoned
};
Nfa build_nfa(const Tree& t, int max_states);   // nfa_node_m.F90:61-106

// ---- Fortran string helpers used by the API layer ------------------------------------------------------
std::string f_trim(const std::string& s);        // TRIM
std::string f_adjustl(const std::string& s);     // ADJUSTL
int f_len_trim(const std::string& s);            // LEN_TRIM
bool f_eq(const std::string& a, const std::string& b);   // == with blank padding
int f_index(const std::string& s, const std::string& sub, bool back = false);   // INDEX (1-based, 0 = none)

}   // namespace fxfe
