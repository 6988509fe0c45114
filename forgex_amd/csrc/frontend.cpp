// See frontend.hpp.  Statement-level restatement of the reference's pattern front end.
#include "frontend.hpp"

#include <algorithm>
#include <cstring>
#include <stdexcept>

namespace fxfe {

namespace {
struct LimitError {
   int code;
};
inline unsigned B(const std::string& s, int i) { return static_cast<unsigned char>(s[static_cast<size_t>(i - 1)]); }
// Fortran substring s(a:b), 1-based inclusive; empty when b < a.
inline std::string sub(const std::string& s, int a, int b) {
   if (b < a) return std::string();
   if (a < 1) a = 1;
   if (b > static_cast<int>(s.size())) b = static_cast<int>(s.size());
   if (b < a) return std::string();
   return s.substr(static_cast<size_t>(a - 1), static_cast<size_t>(b - a + 1));
}
}   // namespace

// ======================================================================================================
// Fortran character semantics
// ======================================================================================================
std::string f_trim(const std::string& s) {
   size_t n = s.size();
   while (n > 0 && s[n - 1] == ' ') --n;
   return s.substr(0, n);
}
int f_len_trim(const std::string& s) { return static_cast<int>(f_trim(s).size()); }
std::string f_adjustl(const std::string& s) {
   size_t k = 0;
   while (k < s.size() && s[k] == ' ') ++k;
   std::string r = s.substr(k);
   r.append(k, ' ');
   return r;
}
bool f_eq(const std::string& a, const std::string& b) {
   size_t n = std::max(a.size(), b.size());
   for (size_t i = 0; i < n; ++i) {
      char ca = i < a.size() ? a[i] : ' ';
      char cb = i < b.size() ? b[i] : ' ';
      if (ca != cb) return false;
   }
   return true;
}
int f_index(const std::string& s, const std::string& subs, bool back) {
   if (subs.empty()) return back ? static_cast<int>(s.size()) + 1 : 1;
   if (subs.size() > s.size()) return 0;
   size_t p = back ? s.rfind(subs) : s.find(subs);
   return p == std::string::npos ? 0 : static_cast<int>(p) + 1;
}

// ======================================================================================================
// status messages (reference src/essential/error_m.F90:40-125)
// ======================================================================================================
const char* status_message(int code) {
   switch (code) {
      case SYNTAX_VALID: return "Given pattern is valid.";
      case SYNTAX_ERR: return "ERROR: Pattern includes some syntax error.";
      case SYNTAX_ERR_PARENTHESIS_MISSING: return "ERROR: Closing parenthesis is expected.";
      case SYNTAX_ERR_PARENTHESIS_UNEXPECTED: return "ERROR: Unexpected closing parenthesis error.";
      case SYNTAX_ERR_BRACKET_MISSING: return "ERROR: Closing square bracket is expected.";
      case SYNTAX_ERR_BRACKET_UNEXPECTED: return "ERROR: Unexpected closing square bracket error.";
      case SYNTAX_ERR_CURLYBRACE_MISSING: return "ERROR: Closing right curlybrace is expected.";
      case SYNTAX_ERR_CURLYBRACE_UNEXPECTED: return "ERROR: Unexpected closing right curlybrace error.";
      case SYNTAX_ERR_INVALID_TIMES: return "ERROR: Given quantifier range is invalid.";
      case SYNTAX_ERR_ESCAPED_SYMBOL_MISSING: return "ERROR: Pattern cannot end with a trailing unescaped backslash.";
      case SYNTAX_ERR_ESCAPED_SYMBOL_INVALID: return "ERROR: This token has no special meaning.";
      case SYNTAX_ERR_EMPTY_CHARACTER_CLASS: return "ERROR: Given class has no character.";
      case SYNTAX_ERR_RANGE_WITH_ESCAPE_SEQUENCES: return "ERROR: Cannot create a range with shorthand escape sequence";
      case SYNTAX_ERR_MISPLACED_SUBTRACTION_OPERATOR:
         return "ERROR: Subtraction operator is misplaced in the given character class.";
      case SYNTAX_ERR_INVALID_CHARACTER_RANGE: return "ERROR: Given character range is invalid.";
      case SYNTAX_ERR_CHAR_CLASS_SUBTRANCTION_NOT_IMPLEMENTED:
         return "ERROR: Character class subtraction hasn't implemented yet.";
      case SYNTAX_ERR_STAR_INCOMPLETE: return "ERROR: Not quantifiable; star '*' operator is missing operand.";
      case SYNTAX_ERR_PLUS_INCOMPLETE: return "ERROR: Not quantifiable; plus '+' operator is missing operand.";
      case SYNTAX_ERR_QUESTION_INCOMPLETE:
         return "ERROR: Not quantifiable; question '?' operator is missing operand.";
      case SYNTAX_ERR_INVALID_HEXADECIMAL:
         return "ERROR: Invalid characters detected. Ensure all characters are 0-9, A-F/a-f.";
      case SYNTAX_ERR_HEX_DIGITS_NOT_ENOUGH:
         return "ERROR: At least 2 hexadecimal digits are required (e.g., '0A' instead of 'A').";
      case SYNTAX_ERR_UNICODE_EXCEED: return "ERROR: Given hex number exceeds the range of unicode codepoint.";
      case ALLOCATION_ERR: return "ERROR: Allocation is failed.";
      case FX_ERR_TREE_LIMIT: return "ERROR: Exceeded the maximum number of tree nodes can be allocated.";
      case FX_ERR_NFA_LIMIT: return "ERROR: NFA exceeds the supported number of states.";
      case FX_ERR_DFA_LIMIT: return "ERROR: Number of DFA states exceeds the limit.";
      case FX_ERR_UNDEFINED: return "ERROR: Pattern has undefined behaviour in the reference implementation.";
      // SYNTAX_ERR_UNICODE_PROPERTY_NOT_IMPLEMENTED has no case in the reference's select -> default branch
      default: return "ERROR: Fatal error is happened.";
   }
}

// ======================================================================================================
// segments (reference src/essential/segment_m.F90)
// ======================================================================================================
const Seg SEG_INIT(UTF8_CODE_MAX + 2, UTF8_CODE_MAX + 2), SEG_ERROR(-2, -2), SEG_EPSILON(-1, -1),
   SEG_EMPTY(UTF8_CODE_EMPTY, UTF8_CODE_EMPTY), SEG_ANY(UTF8_CODE_MIN, UTF8_CODE_MAX), SEG_TAB(9, 9), SEG_LF(10, 10),
   SEG_FF(12, 12), SEG_CR(13, 13), SEG_SPACE(32, 32), SEG_UNDERSCORE(95, 95), SEG_DIGIT(48, 57), SEG_UPPERCASE(65, 90),
   SEG_LOWERCASE(97, 122), SEG_ZENKAKU_SPACE(12288, 12288), SEG_UPPER(UTF8_CODE_MAX + 1, UTF8_CODE_MAX + 1),
   SEG_WHOLE(0, UTF8_CODE_MAX);

bool Seg::validate() const {   // segment_m.F90:185-193
   Seg init;
   return min != init.min && max != init.max && min <= max;
}

static int width_of_segment(const Seg& s) { return s.validate() ? s.max - s.min + 1 : -1; }
static int total_width_of_segment(const std::vector<Seg>& l) {
   int r = 0;
   for (const Seg& s : l) r += width_of_segment(s);
   return r;
}

void sort_segment_by_min(std::vector<Seg>& s) {   // segment_m.F90:450-469 (exchange sort; ties are order-insensitive downstream)
   size_t n = s.size();
   for (size_t i = 0; i + 1 < n; ++i)
      for (size_t j = i + 1; j < n; ++j)
         if (s[i].min > s[j].min) std::swap(s[i], s[j]);
}

void merge_segments(std::vector<Seg>& s) {   // segment_m.F90:472-506
   int n = static_cast<int>(s.size());
   if (n == 0) return;
   int m = 1;
   for (int i = 2; i <= n; ++i) {
      if (s[static_cast<size_t>(i - 1)] == SEG_INIT) break;
      ++m;
   }
   n = m;
   if (n <= 1) {
      s.resize(static_cast<size_t>(n));
      return;
   }
   int j = 1;
   for (int i = 2; i <= n; ++i) {
      Seg& sj = s[static_cast<size_t>(j - 1)];
      const Seg& si = s[static_cast<size_t>(i - 1)];
      if (sj.max >= si.min - 1) {
         sj.max = std::max(sj.max, si.max);
      } else {
         ++j;
         s[static_cast<size_t>(j - 1)] = si;
      }
   }
   if (j <= n) s.resize(static_cast<size_t>(j));
}

void invert_segment_list(std::vector<Seg>& list) {   // segment_m.F90:199-253
   sort_segment_by_min(list);
   merge_segments(list);
   int count = 0;
   int current_min = UTF8_CODE_EMPTY + 1;
   int n = static_cast<int>(list.size());
   for (int i = 0; i < n; ++i) {
      if (current_min < list[static_cast<size_t>(i)].min) ++count;
      current_min = list[static_cast<size_t>(i)].max + 1;
   }
   if (current_min <= UTF8_CODE_MAX) ++count;
   std::vector<Seg> nl(static_cast<size_t>(count));   // default = SEG_INIT, trailing unused entries stay that way
   count = 1;
   current_min = UTF8_CODE_MIN;
   for (int i = 0; i < n; ++i) {
      if (current_min < list[static_cast<size_t>(i)].min) {
         if (count <= static_cast<int>(nl.size())) {
            nl[static_cast<size_t>(count - 1)].min = current_min;
            nl[static_cast<size_t>(count - 1)].max = list[static_cast<size_t>(i)].min - 1;
         }
         ++count;
      }
      current_min = list[static_cast<size_t>(i)].max + 1;
   }
   if (current_min <= UTF8_CODE_MAX && count <= static_cast<int>(nl.size())) {
      nl[static_cast<size_t>(count - 1)].min = current_min;
      nl[static_cast<size_t>(count - 1)].max = UTF8_CODE_MAX;
   }
   list.swap(nl);
}

// hex2seg, segment_m.F90:349-404
static void hex2seg(const std::string& str, Seg& seg, int& ierr) {
   seg = Seg(-1, -1);
   if (f_eq(str, "") || str.size() < 2) {
      ierr = SYNTAX_ERR_HEX_DIGITS_NOT_ENOUGH;
      return;
   }
   // Z<n> edit descriptor: hex digits; blanks inside the field are ignored (BLANK='NULL' default for internal units).
   int64_t v = 0;
   int nd = 0;
   for (char ch : str) {
      int d;
      if (ch == ' ') continue;
      if (ch == ',') break;   // a comma ends a numeric input field early (accepted by the Fortran runtime)
      if (ch >= '0' && ch <= '9') d = ch - '0';
      else if (ch >= 'a' && ch <= 'f') d = ch - 'a' + 10;
      else if (ch >= 'A' && ch <= 'F') d = ch - 'A' + 10;
      else {
         ierr = SYNTAX_ERR_INVALID_HEXADECIMAL;
         return;
      }
      if (v != 0 || d != 0) ++nd;
      if (nd > 8) {   // does not fit the 32-bit target: the runtime reports an input error
         ierr = SYNTAX_ERR_INVALID_HEXADECIMAL;
         return;
      }
      v = v * 16 + d;
   }
   int32_t code = static_cast<int32_t>(static_cast<uint32_t>(v));   // 8 digits with the top bit set read back negative
   if (!(0 <= code && code <= UTF8_CODE_MAX)) {
      ierr = SYNTAX_ERR_UNICODE_EXCEED;
      return;
   }
   seg = Seg(code, code);
   ierr = SYNTAX_VALID;
}

// ======================================================================================================
// UTF-8 (reference src/essential/utf8_m.f90)
// ======================================================================================================
bool is_valid_multiple_byte_character(const std::string& ch) {   // utf8_m.f90:195-246
   int siz = static_cast<int>(ch.size());
   if (siz == 0) return false;
   unsigned b = B(ch, 1);
   int expected;
   if ((b >> 3) == 31) return false;
   else if ((b >> 3) == 30) expected = 4;
   else if ((b >> 4) == 14) expected = 3;
   else if ((b >> 5) == 6) expected = 2;
   else if ((b >> 7) == 0) expected = 1;
   else return false;
   if (expected != siz) return false;
   for (int i = 2; i <= expected; ++i)
      if ((B(ch, i) >> 6) != 2) return false;
   return true;
}

int idxutf8(const std::string& s, int curr) {   // utf8_m.f90:44-140
   int len = static_cast<int>(s.size());
   if (curr > len) return INVALID_CHAR_INDEX;
   int tail = curr;
   for (int i = 0; i <= 3; ++i) {
      if (curr + i > len) return curr;
      unsigned b = B(s, curr + i);
      unsigned s3 = b >> 3, s4 = b >> 4, s5 = b >> 5, s6 = b >> 6, s7 = b >> 7;
      if (s6 == 2) continue;
      if (i == 0) {
         if (s3 == 30) { tail = curr + 3; break; }
         if (s4 == 14) { tail = curr + 2; break; }
         if (s5 == 6) { tail = curr + 1; break; }
         if (s7 == 0) { tail = curr; break; }
      } else {
         if (s3 == 30 || s4 == 14 || s5 == 6 || s7 == 0) { tail = curr + i - 1; break; }
      }
   }
   if (tail <= len) {
      if (!is_valid_multiple_byte_character(sub(s, curr, tail))) tail = curr;
   } else {
      tail = curr;
   }
   return tail;
}

int next_idxutf8(const std::string& s, int curr) {   // utf8_m.f90:146-163
   int e = idxutf8(s, curr);
   return e != INVALID_CHAR_INDEX ? e + 1 : INVALID_CHAR_INDEX;
}

void next_idxutf8_strict(const std::string& s, int curr, int& next, bool& valid) {   // utf8_m.f90:168-191
   valid = false;
   int ie = idxutf8(s, curr);
   if (ie != INVALID_CHAR_INDEX) {
      valid = is_valid_multiple_byte_character(sub(s, curr, ie));
      next = ie + 1;
   } else {
      next = curr + 1;
   }
}

int32_t ichar_utf8(const std::string& ch) {   // utf8_m.f90:338-430
   if (ch.size() > 4) return -1;
   unsigned b[4] = {0, 0, 0, 0};
   for (size_t i = 0; i < ch.size(); ++i) b[i] = static_cast<unsigned char>(ch[i]);
   if (ch.empty()) return 0;
   if ((b[0] >> 7) == 0) return static_cast<int32_t>(b[0]);
   if ((b[0] >> 3) == 30)
      return static_cast<int32_t>(((((((b[0] & 0x07u) << 6) | (b[1] & 0x3Fu)) << 6) | (b[2] & 0x3Fu)) << 6) | (b[3] & 0x3Fu));
   if ((b[0] >> 4) == 14) return static_cast<int32_t>(((((b[0] & 0x0Fu) << 6) | (b[1] & 0x3Fu)) << 6) | (b[2] & 0x3Fu));
   if ((b[0] >> 5) == 6) return static_cast<int32_t>(((b[0] & 0x1Fu) << 6) | (b[1] & 0x3Fu));
   return 0;
}

std::string char_utf8(int32_t code) {   // utf8_m.f90:253-317
   if (!(code > 127)) return std::string(1, static_cast<char>(code & 0xFF));
   unsigned b1 = (static_cast<uint32_t>(code) >> 18) & 0x3Fu, b2 = (static_cast<uint32_t>(code) >> 12) & 0x3Fu,
            b3 = (static_cast<uint32_t>(code) >> 6) & 0x3Fu, b4 = static_cast<uint32_t>(code) & 0x3Fu;
   auto cont = [](unsigned x) { return (x | 0x80u) & ~0x40u; };
   if (code > 65535) {
      b1 = (b1 | 0xF0u) & ~0x08u;
      b2 = cont(b2); b3 = cont(b3); b4 = cont(b4);
   } else if (code > 2047) {
      b1 = 32;
      b2 = (b2 | 0xE0u) & ~0x10u;
      b3 = cont(b3); b4 = cont(b4);
   } else {
      b1 = 32; b2 = 32;
      b3 = (b3 | 0xC0u) & ~0x20u;
      b4 = cont(b4);
   }
   std::string s;
   s.push_back(static_cast<char>(b1 & 0xFF));
   s.push_back(static_cast<char>(b2 & 0xFF));
   s.push_back(static_cast<char>(b3 & 0xFF));
   s.push_back(static_cast<char>(b4 & 0xFF));
   return f_trim(f_adjustl(s));
}

int len_utf8(const std::string& s) {   // utf8_m.f90:466-481
   int i = 1, count = 0, len = static_cast<int>(s.size());
   while (i <= len) {
      int inext = idxutf8(s, i) + 1;
      ++count;
      i = inext;
   }
   return count;
}

std::string reverse_utf8(const std::string& s) {   // utf8_m.f90:596-613
   std::string r;
   int i = 1;
   while (i != INVALID_CHAR_INDEX) {
      int ie = idxutf8(s, i);
      r = sub(s, i, ie) + r;
      i = next_idxutf8(s, i);
   }
   return r;
}

// ======================================================================================================
// tokenizer + parser
// ======================================================================================================
namespace {

enum Token : int {
   tk_char = 0, tk_union, tk_lpar, tk_rpar, tk_backslash, tk_question, tk_star, tk_plus, tk_lsbracket, tk_rsbracket,
   tk_lcurlybrace, tk_rcurlybrace, tk_dot, tk_hyphen, tk_caret, tk_dollar, tk_end
};

std::string pad4(const std::string& s) {
   std::string r = s.substr(0, 4);
   r.append(4 - r.size(), ' ');
   return r;
}

struct CA {   // character_array_t, character_array_m.F90:16-28
   std::string c;
   bool has_c = false;
   bool is_escaped = false, is_hyphenated = false, is_subtract = false;
   int seg_size = 0;
};

struct Parser {
   Tree& t;
   std::string str;
   int idx = 1;
   int current_token = tk_end;
   std::string token_char = std::string(1, '\0') + "   ";   // EMPTY = char(0), syntax_tree_node_m.F90:33,58
   int paren_balance = 0;
   int capacity = 32;

   explicit Parser(Tree& tree) : t(tree) {}

   // ---- tape_t%get_token, syntax_tree_node_m.F90:133-215 ------------------------------------------
   void get_token(bool flag_present = false, bool class_flag = false) {
      int ib = idx;
      if (ib == INVALID_CHAR_INDEX || ib > static_cast<int>(str.size())) {
         current_token = tk_end;
         token_char = "    ";
         return;
      }
      int ie = idxutf8(str, ib);
      std::string c = pad4(sub(str, ib, ie));
      std::string tc = f_trim(c);
      auto is = [&](char sym) { return tc.size() == 1 && tc[0] == sym; };
      if (flag_present) {
         if (class_flag) {
            if (is(']')) current_token = tk_rsbracket;
            else if (is('-')) current_token = tk_hyphen;
            else if (is('\\')) current_token = tk_backslash;
            else current_token = tk_char;
            token_char = c;
         }
      } else {
         if (is('|')) current_token = tk_union;
         else if (is('(')) current_token = tk_lpar;
         else if (is(')')) current_token = tk_rpar;
         else if (is('*')) current_token = tk_star;
         else if (is('+')) current_token = tk_plus;
         else if (is('?')) current_token = tk_question;
         else if (is('\\')) {
            current_token = tk_backslash;
            ib = next_idxutf8(str, ie);
            ie = idxutf8(str, ib);
            token_char = pad4(sub(str, ib, ie));
         } else if (is('[')) current_token = tk_lsbracket;
         else if (is(']')) current_token = tk_rsbracket;
         else if (is('{')) { current_token = tk_lcurlybrace; token_char = c; }
         else if (is('}')) { current_token = tk_rcurlybrace; token_char = c; }
         else if (is('.')) current_token = tk_dot;
         else if (is('^')) current_token = tk_caret;
         else if (is('$')) current_token = tk_dollar;
         else { current_token = tk_char; token_char = c; }
      }
      idx = next_idxutf8(str, ib);
   }

   // ---- node registration, syntax_tree_graph_m.F90:141-199 -------------------------------------------
   int reg(TreeNode node, int left_own, int right_own) {
      int top = t.top + 1;
      if (top > capacity) {
         if (capacity * 2 > TREE_NODE_HARD_LIMIT) throw LimitError{FX_ERR_TREE_LIMIT};
         capacity *= 2;
      }
      if (static_cast<int>(t.nodes.size()) <= top) t.nodes.resize(static_cast<size_t>(top) + 1);
      node.own_i = top;
      t.nodes[static_cast<size_t>(top)] = node;
      t.top = top;
      // connect_left / connect_right
      t.nodes[static_cast<size_t>(top)].left_i = left_own;
      if (left_own != INVALID_INDEX) t.nodes[static_cast<size_t>(left_own)].parent_i = top;
      t.nodes[static_cast<size_t>(top)].right_i = right_own;
      if (right_own != INVALID_INDEX) t.nodes[static_cast<size_t>(right_own)].parent_i = top;
      return top;
   }
   static TreeNode mk(int op) {
      TreeNode n;
      n.op = op;
      return n;
   }
   static TreeNode atom(const Seg& s) {
      TreeNode n;
      n.op = op_char;
      n.c.assign(1, s);
      n.has_c = true;
      return n;
   }
   static constexpr int TERM = INVALID_INDEX;   // `terminal`%own_i

   void fail(int code) {
      t.code = code;
      t.is_valid = false;
   }

   // ---- regex / term / suffix_op / primary, syntax_tree_graph_m.F90:205-443 --------------------------
   void regex() {
      term();
      if (t.is_valid) {
         int left = t.top;
         while (current_token == tk_union) {
            get_token();
            term();
            if (!t.is_valid) break;
            int right = t.top;
            reg(mk(op_union), left, right);
            left = t.top;
         }
      }
   }

   void term() {
      if (current_token == tk_union || current_token == tk_rpar || current_token == tk_end) {
         reg(mk(op_empty), TERM, TERM);
      } else {
         suffix_op();
         if (!t.is_valid) return;
         int left = t.top;
         while (current_token != tk_union && current_token != tk_rpar && current_token != tk_end) {
            suffix_op();
            if (!t.is_valid) return;
            int right = t.top;
            reg(mk(op_concat), left, right);
            left = t.top;
         }
      }
      if (current_token == tk_rpar) paren_balance -= 1;
   }

   void suffix_op() {
      primary();
      if (!t.is_valid) return;
      int left = t.top;
      switch (current_token) {
         case tk_star:
            reg(mk(op_closure), left, TERM);
            get_token();
            break;
         case tk_plus: {
            reg(mk(op_closure), left, TERM);
            int right = t.top;
            reg(mk(op_concat), left, right);
            get_token();
            break;
         }
         case tk_question: {
            reg(mk(op_empty), left, TERM);
            int right = t.top;
            reg(mk(op_union), left, right);
            get_token();
            break;
         }
         case tk_lcurlybrace:
            times();
            if (!t.is_valid) return;
            get_token();
            break;
         default: break;
      }
   }

   void primary() {
      switch (current_token) {
         case tk_char: {
            int32_t code = ichar_utf8(token_char);
            reg(atom(Seg(code, code)), TERM, TERM);
            get_token();
            break;
         }
         case tk_lpar:
            paren_balance += 1;
            get_token();
            regex();
            if (!t.is_valid) return;
            if (current_token != tk_rpar) { fail(SYNTAX_ERR_PARENTHESIS_MISSING); return; }
            get_token();
            break;
         case tk_lsbracket:
            char_class();
            if (!t.is_valid) return;
            if (current_token != tk_rsbracket) { fail(SYNTAX_ERR_BRACKET_MISSING); return; }
            get_token();
            break;
         case tk_backslash:
            shorthand();
            if (!t.is_valid) return;
            get_token();
            break;
         case tk_dot:
            reg(atom(SEG_ANY), TERM, TERM);
            get_token();
            break;
         case tk_caret:
         case tk_dollar:
            caret_dollar();
            get_token();
            break;
         case tk_rsbracket: fail(SYNTAX_ERR_BRACKET_UNEXPECTED); return;
         case tk_rpar: fail(SYNTAX_ERR_PARENTHESIS_UNEXPECTED); return;
         case tk_rcurlybrace: {
            int32_t code = ichar_utf8(token_char);
            reg(atom(Seg(code, code)), TERM, TERM);
            get_token();
            break;
         }
         case tk_lcurlybrace: fail(SYNTAX_ERR_INVALID_TIMES); return;
         case tk_star: fail(SYNTAX_ERR_STAR_INCOMPLETE); return;
         case tk_plus: fail(SYNTAX_ERR_PLUS_INCOMPLETE); return;
         case tk_question: fail(SYNTAX_ERR_QUESTION_INCOMPLETE); return;
         default: fail(SYNTAX_ERR_THIS_SHOULD_NOT_HAPPEN); return;
      }
   }

   // ---- crlf / caret_dollar, syntax_tree_graph_m.F90:559-605 -------------------------------------------
   void crlf() {
      int cr = reg(atom(SEG_CR), TERM, TERM);
      int lf = reg(atom(SEG_LF), TERM, TERM);
      int right = reg(mk(op_concat), cr, lf);
      reg(mk(op_union), lf, right);
   }
   void caret_dollar() {
      int cr = reg(atom(SEG_CR), TERM, TERM);
      int lf = reg(atom(SEG_LF), TERM, TERM);
      int node_r_r = reg(mk(op_concat), cr, lf);
      int node_r = reg(mk(op_union), lf, node_r_r);
      int empty_r = reg(atom(SEG_EMPTY), TERM, TERM);
      reg(mk(op_union), node_r, empty_r);
   }

   // ---- shorthand, syntax_tree_graph_m.F90:611-725 -------------------------------------------------------
   void shorthand() {
      std::string tc = f_trim(token_char);
      std::vector<Seg> seglist;
      auto is = [&](char sym) { return tc.size() == 1 && tc[0] == sym; };
      if (is('t')) { reg(atom(SEG_TAB), TERM, TERM); return; }
      if (is('n')) { crlf(); return; }
      if (is('r')) { reg(atom(SEG_CR), TERM, TERM); return; }
      if (is('d')) { reg(atom(SEG_DIGIT), TERM, TERM); return; }
      if (is('D')) {
         seglist = {SEG_DIGIT};
         invert_segment_list(seglist);
      } else if (is('w')) {
         seglist = {SEG_LOWERCASE, SEG_UPPERCASE, SEG_DIGIT, SEG_UNDERSCORE};
      } else if (is('W')) {
         seglist = {SEG_LOWERCASE, SEG_UPPERCASE, SEG_DIGIT, SEG_UNDERSCORE};
         invert_segment_list(seglist);
      } else if (is('s')) {
         seglist = {SEG_SPACE, SEG_TAB, SEG_CR, SEG_LF, SEG_FF, SEG_ZENKAKU_SPACE};
      } else if (is('S')) {
         seglist = {SEG_SPACE, SEG_TAB, SEG_CR, SEG_LF, SEG_FF, SEG_ZENKAKU_SPACE};
         invert_segment_list(seglist);
      } else if (is('x')) {
         hexadecimal_to_segment(seglist);
         if (!t.is_valid) return;
      } else if (tc.empty()) {
         fail(SYNTAX_ERR_ESCAPED_SYMBOL_MISSING);
         return;
      } else if (tc.size() == 1 && std::strchr("[]{}()$\\|.?^*+-", tc[0]) != nullptr) {
         int32_t code = ichar_utf8(token_char);
         reg(atom(Seg(code, code)), TERM, TERM);
         return;
      } else {
         fail(SYNTAX_ERR_ESCAPED_SYMBOL_INVALID);
         return;
      }
      TreeNode node;
      node.op = op_char;
      node.c = seglist;
      node.has_c = true;
      reg(node, TERM, TERM);
   }

   // ---- \x.. and \x{...}, syntax_tree_graph_m.F90:728-777 ---------------------------------------------------
   void hexadecimal_to_segment(std::vector<Seg>& seglist) {
      std::string hex;
      get_token();
      bool is_longer_digit = current_token == tk_lcurlybrace;
      bool is_two_digit = !is_longer_digit;
      if (is_longer_digit) get_token();
      hex = token_char.substr(0, 1);
      int i = 2;
      while (true) {
         if (is_two_digit && i >= 3) break;
         get_token();
         if (is_longer_digit && current_token != tk_rcurlybrace && current_token != tk_char) {
            t.is_valid = false;
            t.code = SYNTAX_ERR_CURLYBRACE_MISSING;
            return;
         }
         if (current_token == tk_rcurlybrace) break;
         hex += token_char.substr(0, 1);
         ++i;
      }
      seglist.assign(1, Seg());
      hex2seg(f_trim(hex), seglist[0], t.code);
      if (t.code != SYNTAX_VALID) {
         t.is_valid = false;
         return;
      }
      t.is_valid = SEG_WHOLE.min <= seglist[0].min && seglist[0].max <= SEG_WHOLE.max;
      if (!t.is_valid) t.code = SYNTAX_ERR_UNICODE_EXCEED;
   }

   // ---- {m,n}, syntax_tree_graph_m.F90:782-906 ---------------------------------------------------------------
   // list-directed READ of one integer from an internal unit: 0 ok, >0 error, <0 end of file
   static int list_read_int(const std::string& s, int& val) {
      size_t p = 0;
      while (p < s.size() && s[p] == ' ') ++p;
      if (p >= s.size()) return -1;
      if (s[p] == ',' ) return 0;    // null value: item keeps its definition status
      if (s[p] == '/') return 0;     // slash terminates the list
      size_t q = p;
      bool neg = false;
      if (s[q] == '+' || s[q] == '-') { neg = s[q] == '-'; ++q; }
      size_t d0 = q;
      int64_t v = 0;
      while (q < s.size() && s[q] >= '0' && s[q] <= '9') {
         v = v * 10 + (s[q] - '0');
         if (v > 4294967296LL) return 1;
         ++q;
      }
      if (q == d0) return 1;
      if (q < s.size() && s[q] != ' ' && s[q] != ',' && s[q] != '/') return 1;   // includes r*c forms, decimals, letters
      if (neg) v = -v;
      if (v > 2147483647LL || v < -2147483648LL) return 1;
      val = static_cast<int>(v);
      return 0;
   }
   static bool is_integer(const std::string& chara) {   // utility_m.f90:146-169
      if (chara.find(',') != std::string::npos || chara.find(' ') != std::string::npos) return false;
      std::string f = chara.substr(0, 19);
      size_t q = 0;
      if (q < f.size() && (f[q] == '+' || f[q] == '-')) ++q;
      if (q >= f.size()) return false;
      for (; q < f.size(); ++q)
         if (f[q] < '0' || f[q] > '9') return false;
      return true;
   }

   void times() {
      std::string buf;
      int arg1 = INVALID_REPEAT_VAL, arg2 = INVALID_REPEAT_VAL;
      bool is_infinite = false;
      int mx = INVALID_REPEAT_VAL, mn = INVALID_REPEAT_VAL;
      get_token();
      while (current_token != tk_rcurlybrace) {
         buf += f_trim(token_char);
         get_token();
         if (current_token == tk_end) { fail(SYNTAX_ERR_CURLYBRACE_MISSING); return; }
      }
      if (buf.empty()) { fail(SYNTAX_ERR_INVALID_TIMES); return; }
      if (buf.size() == 1 && buf[0] == ',') { fail(SYNTAX_ERR_INVALID_TIMES); return; }
      if (buf[0] == ',') buf = "0" + buf;
      if (is_integer(buf)) buf = f_trim(buf) + "," + f_trim(buf);
      // get_index_comma, utility_m.f90:122-142
      int i = 0, num_comma = 0;
      for (size_t k = 0; k < buf.size(); ++k)
         if (buf[k] == ',') {
            if (i == 0) i = static_cast<int>(k) + 1;
            ++num_comma;
         }
      if (num_comma > 1) { fail(SYNTAX_ERR_INVALID_TIMES); return; }
      std::string c1 = sub(buf, 1, i - 1), c2;
      if (i + 1 <= f_len_trim(buf)) c2 = sub(buf, i + 1, f_len_trim(buf));
      int ios = list_read_int(c1, arg1);
      if (ios > 0 || arg1 < 0) { fail(SYNTAX_ERR_INVALID_TIMES); return; }
      if (f_trim(c2).empty()) {
         is_infinite = true;
      } else {
         ios = list_read_int(c2, arg2);
         if (ios > 0 || arg2 < 0) { fail(SYNTAX_ERR_INVALID_TIMES); return; }
      }
      if (is_infinite) { mn = arg1; mx = INFINITE_REPEAT; }
      else { mn = arg1; mx = arg2; }
      if (mn == 0 && mx == 0) {
      } else if (mx != INFINITE_REPEAT && mn > mx) {
         fail(SYNTAX_ERR_INVALID_TIMES);
         return;
      }
      TreeNode node;
      node.op = op_repeat;
      node.min_repeat = mn;
      node.max_repeat = mx;
      int left = t.top;
      reg(node, left, TERM);
   }

   // ---- [...] , syntax_tree_graph_m.F90:448-556 ------------------------------------------------------------------
   void char_class() {
      get_token(true, true);
      std::string buf, curr;
      bool backslashed = false;
      while (current_token != tk_rsbracket) {
         if (current_token == tk_end) return;
         int ie = idxutf8(token_char, 1);
         curr = sub(token_char, 1, ie);
         buf += curr;
         backslashed = (current_token == tk_backslash && !backslashed);
         get_token(true, true);
         if (current_token == tk_rsbracket && backslashed) {
            ie = idxutf8(token_char, 1);
            curr = sub(token_char, 1, ie);
            buf += curr;
            get_token(true, true);
         }
      }
      if (buf.empty()) { fail(SYNTAX_ERR_EMPTY_CHARACTER_CLASS); return; }
      bool is_inverted = false;
      if (buf[0] == '^') {
         is_inverted = true;
         buf = buf.substr(1);
      }
      if (len_utf8(buf) < 1) { fail(SYNTAX_ERR_EMPTY_CHARACTER_CLASS); return; }
      std::vector<Seg> seglist;
      bool allocated = false;
      interpret_class_string(buf, seglist, allocated, t.is_valid, t.code);
      if (!t.is_valid) return;
      if (!allocated) { fail(ALLOCATION_ERR); return; }
      if (seglist.empty()) { fail(SYNTAX_ERR_THIS_SHOULD_NOT_HAPPEN); return; }
      if (is_inverted) invert_segment_list(seglist);
      TreeNode node;
      node.op = op_char;
      node.c = seglist;
      node.has_c = true;
      reg(node, TERM, TERM);
   }

   // character_array_m.F90:75-140
   static void parse_backslash_and_hyphen(std::vector<CA>& array, int& ierr) {
      int n = static_cast<int>(array.size());
      if (n < 1) return;
      std::vector<CA> temp(static_cast<size_t>(n) + 2);   // 1-based, slot 0 absorbs the reference's out-of-range write
      int k = 1;
      bool zone = false;
      auto A = [&](int i) -> CA& { return array[static_cast<size_t>(i - 1)]; };
      for (int i = 1; i <= n; ++i) {
         if (1 < i && i < n) {
            if (!zone) {
               if (A(i).c == "-" && A(i + 1).c == "-") {
                  for (int q = k; q <= n; ++q) temp[static_cast<size_t>(q)].is_subtract = true;
                  zone = true;
                  continue;
               }
            } else {
               if (A(i).c == "-" && A(i + 1).c == "-") {
                  ierr = SYNTAX_ERR_MISPLACED_SUBTRACTION_OPERATOR;
                  return;
               }
            }
            if (A(i - 1).c == "-" && A(i).c == "-") continue;
         }
         if (A(i).c == "\\" && !temp[static_cast<size_t>(k)].is_escaped) {
            temp[static_cast<size_t>(k)].is_escaped = true;
         } else if (A(i).c == "-" && i != 1) {
            temp[static_cast<size_t>(k - 1)].is_hyphenated = true;   // k == 1 writes before the array in the reference
         } else {
            temp[static_cast<size_t>(k)].c = A(i).c;
            temp[static_cast<size_t>(k)].has_c = true;
            ++k;
         }
      }
      int siz = k - 1;
      array.assign(temp.begin() + 1, temp.begin() + 1 + siz);
   }

   static bool in_hex(const CA& e) {
      int32_t code = ichar_utf8(e.c);
      return (48 <= code && code <= 57) || (65 <= code && code <= 70) || (97 <= code && code <= 102);
   }

   // character_array_m.F90:225-332
   static void parse_escape_sequence_with_argument(std::vector<CA>& ca, int& ierr) {
      ierr = SYNTAX_VALID;
      int siz = static_cast<int>(ca.size());
      if (siz == 0) throw LimitError{FX_ERR_UNDEFINED};   // reference copies tmp(1:1) out of a zero-sized array
      std::vector<CA> tmp(static_cast<size_t>(siz) + 2);
      std::string hex_long;
      auto C = [&](int i) -> CA& { return ca[static_cast<size_t>(i - 1)]; };
      int k = 1, j = 1;
      while (j <= siz) {
         if (C(j).c == "x" && C(j).is_escaped) {
            tmp[static_cast<size_t>(k)].c = "x";
            tmp[static_cast<size_t>(k)].has_c = true;
            tmp[static_cast<size_t>(k)].is_escaped = true;
            ++j;
            if (j > siz) break;
            ++k;
            if (j + 1 <= siz) {
               if (in_hex(C(j)) && in_hex(C(j + 1))) {
                  std::string two = (f_trim(C(j).c) + f_trim(C(j + 1).c)).substr(0, 2);
                  two.append(2 - two.size(), ' ');
                  tmp[static_cast<size_t>(k)].c = f_trim(f_adjustl(two));
                  tmp[static_cast<size_t>(k)].has_c = true;
                  tmp[static_cast<size_t>(k)].is_hyphenated = C(j + 1).is_hyphenated;
                  j += 2;
                  if (j > siz) break;
                  ++k;
                  continue;
               } else if (C(j).c == "{") {
                  int i = j + 1;
                  while (true) {
                     if (i > siz) { ierr = SYNTAX_ERR_CURLYBRACE_MISSING; return; }
                     if (C(i).c != "}" && !in_hex(C(i))) { ierr = SYNTAX_ERR_INVALID_HEXADECIMAL; return; }
                     else if (C(i).c == "}") break;
                     hex_long = f_trim(f_adjustl(hex_long)) + C(i).c;
                     ++i;
                  }
                  tmp[static_cast<size_t>(k)].c = f_trim(f_adjustl(hex_long));
                  tmp[static_cast<size_t>(k)].has_c = true;
                  tmp[static_cast<size_t>(k)].is_hyphenated = C(i).is_hyphenated;
                  j = i + 1;
                  if (j > siz) break;
                  ++k;
                  hex_long.clear();
                  continue;
               } else {
                  ierr = SYNTAX_ERR_INVALID_HEXADECIMAL;
                  return;
               }
            } else {
               ierr = SYNTAX_ERR_HEX_DIGITS_NOT_ENOUGH;
               return;
            }
         } else if (C(j).c == "p") {
            ierr = SYNTAX_ERR_UNICODE_PROPERTY_NOT_IMPLEMENTED;
            return;
         }
         tmp[static_cast<size_t>(k)] = C(j);
         ++j;
         if (j > siz) break;
         ++k;
      }
      ca.assign(tmp.begin() + 1, tmp.begin() + 1 + k);
   }

   // character_array_m.F90:145-222
   static void parse_segment_width(std::vector<CA>& array) {
      for (CA& e : array) {
         int n;
         if (e.is_escaped) {
            const std::string& c = e.c;
            auto is = [&](const char* s) { return f_eq(c, s); };
            if (is("t")) n = 1;
            else if (is("n")) n = 2;
            else if (is("r")) n = 1;
            else if (is("d")) n = 10;
            else if (is("D")) { std::vector<Seg> s{SEG_DIGIT}; invert_segment_list(s); n = total_width_of_segment(s); }
            else if (is("w")) { std::vector<Seg> s{SEG_LOWERCASE, SEG_UPPERCASE, SEG_DIGIT, SEG_UNDERSCORE}; n = total_width_of_segment(s); }
            else if (is("W")) { std::vector<Seg> s{SEG_LOWERCASE, SEG_UPPERCASE, SEG_DIGIT, SEG_UNDERSCORE}; invert_segment_list(s); n = total_width_of_segment(s); }
            else if (is("s")) n = 6;
            else if (is("S")) { std::vector<Seg> s{SEG_SPACE, SEG_TAB, SEG_CR, SEG_LF, SEG_FF, SEG_ZENKAKU_SPACE}; invert_segment_list(s); n = total_width_of_segment(s); }
            else if (is("x") || is("\\") || is("{") || is("}") || is("[") || is("]")) n = 1;
            else n = -1;
         } else {
            n = 1;
         }
         e.seg_size = n;
      }
   }

   // syntax_tree_graph_m.F90:1123-1213
   static void convert_escaped(const std::string& chara, std::vector<Seg>& out) {
      std::string tc = f_trim(chara);
      auto is = [&](char s) { return tc.size() == 1 && tc[0] == s; };
      out.clear();
      if (is('t')) out = {SEG_TAB};
      else if (is('n')) out = {SEG_LF, SEG_CR};
      else if (is('r')) out = {SEG_CR};
      else if (is('d')) out = {SEG_DIGIT};
      else if (is('D')) { out = {SEG_DIGIT}; invert_segment_list(out); }
      else if (is('w')) out = {SEG_LOWERCASE, SEG_UPPERCASE, SEG_DIGIT, SEG_UNDERSCORE};
      else if (is('W')) { out = {SEG_LOWERCASE, SEG_UPPERCASE, SEG_DIGIT, SEG_UNDERSCORE}; invert_segment_list(out); }
      else if (is('s')) out = {SEG_SPACE, SEG_TAB, SEG_CR, SEG_LF, SEG_FF, SEG_ZENKAKU_SPACE};
      else if (is('S')) { out = {SEG_SPACE, SEG_TAB, SEG_CR, SEG_LF, SEG_FF, SEG_ZENKAKU_SPACE}; invert_segment_list(out); }
      else if (is('x')) { out.assign(1, Seg()); int unused = 0; hex2seg(chara, out[0], unused); }
      else if (is('p')) out = {SEG_ERROR};
      else if (is('\\') || is('{') || is('}') || is('[') || is(']')) out = {Seg(static_cast<unsigned char>(tc[0]), static_cast<unsigned char>(tc[0]))};
      else out = {SEG_ERROR};
   }

   // register_segment_to_list, segment_m.F90:327-344
   static void reg_seg(std::vector<Seg>& list, const Seg& seg, int& k, int& ierr) {
      if (seg.validate() && k <= static_cast<int>(list.size()) - 1) {
         ++k;
         list[static_cast<size_t>(k - 1)] = seg;
         ierr = 0;
      } else {
         ierr = 1;
      }
   }

   // syntax_tree_graph_m.F90:910-1118
   static void interpret_class_string(const std::string& str, std::vector<Seg>& seglist, bool& allocated, bool& is_valid,
                                      int& ierr) {
      is_valid = true;
      allocated = false;
      bool backslashed = false, prev_hyphenated = false, curr_hyphenated = false;
      Seg prev_seg, curr_seg;
      if (str.size() >= 2 && str.compare(0, 2, "--") == 0) {
         ierr = SYNTAX_ERR_MISPLACED_SUBTRACTION_OPERATOR;
         is_valid = false;
      }
      // character_string_to_array, character_array_m.F90:45-70
      std::vector<CA> ca;
      {
         int siz = len_utf8(str);
         if (siz < 1) {
            ierr = SYNTAX_ERR_EMPTY_CHARACTER_CLASS;
            is_valid = false;
            return;
         }
         ca.resize(static_cast<size_t>(siz));
         int ib = 0, ie = 0;
         for (int j = 1; j <= siz; ++j) {
            ib = ie + 1;
            ie = idxutf8(str, ib);
            if (ib == INVALID_CHAR_INDEX || ie == INVALID_CHAR_INDEX) break;
            ca[static_cast<size_t>(j - 1)].c = sub(str, ib, ie);
            ca[static_cast<size_t>(j - 1)].has_c = true;
         }
      }
      parse_backslash_and_hyphen(ca, ierr);
      if (ierr == SYNTAX_ERR_MISPLACED_SUBTRACTION_OPERATOR) {
         is_valid = false;
         return;
      }
      parse_escape_sequence_with_argument(ca, ierr);
      if (ierr != SYNTAX_VALID) {
         is_valid = false;
         return;
      }
      parse_segment_width(ca);

      int siz = 0;
      for (int i = 1; i <= static_cast<int>(ca.size()); ++i) {
         CA& e = ca[static_cast<size_t>(i - 1)];
         if (e.is_hyphenated && e.seg_size != 1) {
            ierr = SYNTAX_ERR_RANGE_WITH_ESCAPE_SEQUENCES;
            is_valid = false;
            return;
         }
         if (i > 1 && ca[static_cast<size_t>(i - 2)].is_hyphenated && e.seg_size != 1) {
            ierr = SYNTAX_ERR_RANGE_WITH_ESCAPE_SEQUENCES;
            is_valid = false;
            return;
         }
         if (e.is_subtract) {
            ierr = SYNTAX_ERR_CHAR_CLASS_SUBTRANCTION_NOT_IMPLEMENTED;
            is_valid = false;
            return;
         }
         if (i > 1 && i == static_cast<int>(ca.size())) {
            if (e.is_hyphenated) {
               e.is_hyphenated = false;
               CA h;
               h.c = "-";
               h.has_c = true;
               h.is_subtract = e.is_subtract;
               h.seg_size = 1;
               ca.push_back(h);
               siz += 1;
               break;
            }
         }
         siz += e.seg_size;
      }
      if (siz < 1) {
         ierr = SYNTAX_ERR_THIS_SHOULD_NOT_HAPPEN;
         is_valid = false;
         return;
      }
      std::vector<Seg> list(static_cast<size_t>(siz));
      int j = 0;
      int i = 1;
      std::vector<Seg> cache;
      while (i <= static_cast<int>(ca.size())) {
         std::string c = ca[static_cast<size_t>(i - 1)].c;
         backslashed = ca[static_cast<size_t>(i - 1)].is_escaped;
         curr_hyphenated = ca[static_cast<size_t>(i - 1)].is_hyphenated;
         if (i > 1) prev_hyphenated = ca[static_cast<size_t>(i - 2)].is_hyphenated;
         if (backslashed && f_eq(c, "x")) {
            ++i;
            if (i > static_cast<int>(ca.size())) {
               ierr = SYNTAX_ERR_THIS_SHOULD_NOT_HAPPEN;
               is_valid = false;
               return;
            }
            c = ca[static_cast<size_t>(i - 1)].c;
            backslashed = ca[static_cast<size_t>(i - 1)].is_escaped;
            hex2seg(c, curr_seg, ierr);
            if (ierr != SYNTAX_VALID) {
               is_valid = false;
               return;
            }
         } else if (backslashed && f_eq(c, "p")) {
            ierr = SYNTAX_ERR_UNICODE_PROPERTY_NOT_IMPLEMENTED;
            is_valid = false;
            return;
         } else {
            curr_seg = Seg(ichar_utf8(c), ichar_utf8(c));
         }
         if (backslashed) {
            convert_escaped(c, cache);
            if (cache[0] == SEG_ERROR) {
               ierr = SYNTAX_ERR_ESCAPED_SYMBOL_INVALID;
               is_valid = false;
               return;
            }
            if (cache.size() > 1) {
               for (const Seg& s : cache) reg_seg(list, s, j, ierr);
               prev_seg = Seg();
               ++i;
               continue;
            }
            curr_seg = cache[0];
         }
         if (prev_hyphenated) {
            // join_two_segments, segment_m.F90:436-447
            Seg joined(prev_seg.min, curr_seg.max);
            if (!joined.validate()) joined = SEG_INIT;
            curr_seg = joined;
            if (curr_seg == SEG_ERROR) {
               ierr = SYNTAX_ERR_THIS_SHOULD_NOT_HAPPEN;
               is_valid = false;
               return;
            }
         }
         if (!curr_hyphenated) {
            int jerr = 0;
            reg_seg(list, curr_seg, j, jerr);
            if (jerr == 1) {
               ierr = SYNTAX_ERR_INVALID_CHARACTER_RANGE;
               is_valid = false;
               return;
            }
         }
         prev_seg = curr_seg;
         ++i;
      }
      if (j < 1) {
         ierr = SYNTAX_ERR_THIS_SHOULD_NOT_HAPPEN;
         is_valid = false;
         return;
      }
      seglist.assign(list.begin(), list.begin() + j);
      allocated = true;
   }
};

}   // namespace

void Tree::build(const std::string& pattern) {   // syntax_tree_graph_m.F90:61-95
   nodes.assign(1, TreeNode());
   top = 0;
   is_valid = true;
   code = SYNTAX_VALID;
   Parser p(*this);
   p.str = pattern;
   p.idx = 1;
   try {
      p.get_token();
      p.regex();
   } catch (const LimitError& e) {
      is_valid = false;
      code = e.code;
      return;
   }
   if (!is_valid) return;
   if (p.paren_balance > 0) {
      is_valid = false;
      code = SYNTAX_ERR_PARENTHESIS_MISSING;
   } else if (p.paren_balance < 0) {
      is_valid = false;
      code = SYNTAX_ERR_PARENTHESIS_UNEXPECTED;
   }
   if (top >= 1) nodes[static_cast<size_t>(top)].parent_i = 0;
}

// ======================================================================================================
// literal extraction (reference src/ast/syntax_tree_optimize_m.F90)
// ======================================================================================================
namespace {
struct Lit {
   std::string all, pref, suff, fact;
   bool flag_closure = false, flag_class = false;
};

std::string best(const std::string& c1, const std::string& c2) {   // :229-241
   return f_len_trim(c1) > f_len_trim(c2) ? f_trim(f_adjustl(c1)) : f_trim(f_adjustl(c2));
}

std::string same_part_of_prefix(const std::string& c1, const std::string& c2) {   // :244-273
   std::string res;
   int i = 1;
   while (true) {
      std::string part1 = sub(c1, i, idxutf8(c1, i));
      std::string part2 = sub(c2, i, idxutf8(c2, i));
      bool flag_return = next_idxutf8(c1, i) == INVALID_CHAR_INDEX || next_idxutf8(c2, i) == INVALID_CHAR_INDEX;
      if (flag_return) return res;
      if (f_eq(part1, part2)) res += part1;
      else break;
      i = next_idxutf8(c1, i);
   }
   return res;
}

std::string same_part_of_suffix(const std::string& c1, const std::string& c2) {   // :276-291
   return reverse_utf8(same_part_of_prefix(reverse_utf8(c1), reverse_utf8(c2)));
}

void best_factor(const std::vector<TreeNode>& nodes, int idx, Lit& lit) {   // :71-226
   const TreeNode& curr = nodes[static_cast<size_t>(idx)];
   Lit lit_l, lit_r;
   lit.all.clear();
   lit.pref.clear();
   lit.suff.clear();
   lit.fact.clear();
   if (curr.op == op_union || curr.op == op_concat) {
      best_factor(nodes, curr.left_i, lit_l);
      best_factor(nodes, curr.right_i, lit_r);
   }
   switch (curr.op) {
      case op_union:
         lit.pref = same_part_of_prefix(lit_l.pref, lit_r.pref);
         lit.suff = same_part_of_suffix(lit_l.suff, lit_r.suff);
         lit.flag_closure = true;
         break;
      case op_concat: {
         lit.flag_class = lit_l.flag_class || lit_r.flag_class;
         lit.flag_closure = lit_l.flag_closure || lit_r.flag_closure;
         bool Lc = lit_l.flag_class, Rc = lit_r.flag_class, Lk = lit_l.flag_closure, Rk = lit_r.flag_closure;
         if (!Lc && !Rc) {
            if (!Lk && !Rk) {
               lit.all = lit_l.all + lit_r.all;
               lit.pref = best(lit_l.pref, lit_l.all + lit_r.pref);
               lit.suff = best(lit_r.suff, lit_l.suff + lit_r.all);
            } else if (!Lk && Rk) {
               lit.pref = lit_l.all + lit_r.pref;
               lit.suff = lit_r.suff;
            } else if (Lk && !Rk) {
               lit.pref = lit_l.pref;
               lit.suff = lit_l.suff + lit_r.all;
            } else {
               lit.pref = lit_l.pref;
               lit.suff = lit_r.suff;
            }
         } else if (!Lc && Rc) {
            if (!Lk) {   // R_class_N_closure and R_class_R_closure
               lit.pref = best(lit_l.pref, lit_l.all + lit_r.pref);
               lit.suff = lit_r.suff;
            } else {
               lit.pref = lit_l.pref;
               lit.suff = lit_r.suff;
            }
         } else if (Lc && !Rc) {
            if (!Rk) {   // L_class_N_closure and L_class_L_closure
               lit.pref = lit_l.pref;
               lit.suff = best(lit_r.suff, lit_l.suff + lit_r.all);
            } else {
               lit.pref = lit_l.pref;
               lit.suff = lit_r.suff;
            }
         } else {
            if (!Lk && Rk) {   // LR_class_R_closure assigns pref twice and never suff
               lit.pref = lit_l.pref;
            } else {
               lit.pref = lit_l.pref;
               lit.suff = lit_r.suff;
            }
         }
         break;
      }
      case op_closure: lit.flag_closure = true; break;
      case op_char:
         if (curr.has_c) {
            if (curr.c.size() == 1) {
               if (width_of_segment(curr.c[0]) == 1) {
                  lit.all = lit.pref = lit.suff = lit.fact = char_utf8(curr.c[0].min);
               } else {
                  lit.flag_class = true;
               }
            } else {
               lit.flag_class = true;
            }
         }
         break;
      case op_repeat: {
         best_factor(nodes, curr.left_i, lit_l);
         lit.flag_class = lit_l.flag_class;
         for (int i = 1; i <= curr.min_repeat; ++i) {
            best_factor(nodes, curr.left_i, lit_l);
            lit.all += lit_l.all;
            lit.pref += lit_l.pref;
            lit.suff += lit_l.suff;
            lit.fact += lit_l.fact;
            lit.flag_class = lit.flag_class || lit_l.flag_class;
            if (lit_l.flag_closure) break;
         }
         lit.flag_closure = curr.min_repeat != curr.max_repeat;
         lit.flag_closure = lit.flag_closure || lit_l.flag_closure;
         break;
      }
      default: lit.flag_closure = true; break;
   }
}
}   // namespace

Literals extract_literal(const Tree& t) {
   Lit lit;
   best_factor(t.nodes, t.top, lit);
   Literals r;
   r.all = lit.all;
   r.prefix = lit.pref;
   r.suffix = lit.suff;
   return r;
}

// ======================================================================================================
// NFA (reference src/nfa/nfa_node_m.F90, src/essential/segment_disjoin_m.F90)
// ======================================================================================================
bool NfaTransition::is_epsilon() const {
   for (const Seg& s : c)
      if (s == SEG_EPSILON) return true;
   return false;
}
bool NfaTransition::accepts(int32_t code) const {
   for (const Seg& s : c)
      if (s.min <= code && code <= s.max) return true;
   return false;
}

void disjoin(std::vector<Seg>& list) {   // segment_disjoin_m.F90:36-182 (the heap only sorts; membership is order-free)
   int siz = static_cast<int>(list.size());
   if (siz <= 0) return;
   std::vector<Seg> buff = list;
   std::vector<int32_t> index_list;
   index_list.reserve(static_cast<size_t>(siz) * 6);
   for (const Seg& s : buff) {
      index_list.push_back(s.min - 1);
      index_list.push_back(s.min);
      index_list.push_back(s.min + 1);
      index_list.push_back(s.max - 1);
      index_list.push_back(s.max);
      index_list.push_back(s.max + 1);
   }
   std::sort(index_list.begin(), index_list.end());
   index_list.erase(std::unique(index_list.begin(), index_list.end()), index_list.end());
   std::vector<Seg> out;
   Seg nw = SEG_UPPER;
   auto reg = [&](void) {
      if (nw.validate()) out.push_back(nw);
      nw = SEG_UPPER;
   };
   for (int32_t i : index_list) {
      bool in_any = false;
      for (const Seg& s : buff)
         if (s.min <= i && i <= s.max) { in_any = true; break; }
      if (!in_any) continue;
      if (i < nw.min) nw.min = i;
      bool flag = false;
      for (const Seg& s : buff)
         if (i + 1 == s.min) flag = true;
      if (flag) {
         nw.max = i;
         reg();
         continue;
      }
      int count = 0;
      for (const Seg& s : buff)
         if (s.min == i) ++count;
      if (count > 1) {
         nw.max = i;
         reg();
      }
      count = 0;
      for (const Seg& s : buff)
         if (s.max == i) ++count;
      if (count > 0) {
         nw.max = i;
         reg();
      }
   }
   list.swap(out);
}

namespace {
struct BuildTra {   // nfa_transition_t with the reference's fixed-size segment array
   std::vector<Seg> c;
   int c_top = 0;
   int dst = NFA_NULL_TRANSITION;
};
struct BuildNode {
   std::vector<BuildTra> forward;   // registered transitions; reference slot forward_top == forward.size()+1
};
struct NfaBuilder {
   const Tree& tree;
   std::vector<BuildNode> g;   // 1-based
   int nfa_top = 0;
   int max_states;

   NfaBuilder(const Tree& t, int mx) : tree(t), g(1), max_states(mx) {}

   int make_node() {
      ++nfa_top;
      if (nfa_top > max_states) throw LimitError{FX_ERR_NFA_LIMIT};
      if (static_cast<int>(g.size()) <= nfa_top) g.resize(static_cast<size_t>(nfa_top) + 1);
      return nfa_top;
   }

   void add_transition(int src, int dst, const Seg& c) {   // nfa_node_m.F90:324-372 (forward half)
      BuildNode& self = g[static_cast<size_t>(src)];
      int j = -1;
      if (!self.forward.empty() && c != SEG_EPSILON) {
         for (size_t jj = 0; jj < self.forward.size(); ++jj)
            if (dst == self.forward[jj].dst && self.forward[jj].c_top < NFA_C_SIZE) j = static_cast<int>(jj);
      }
      if (j < 0) {
         self.forward.emplace_back();
         j = static_cast<int>(self.forward.size()) - 1;
         self.forward[static_cast<size_t>(j)].c.assign(NFA_C_SIZE, SEG_INIT);
      }
      BuildTra& tr = self.forward[static_cast<size_t>(j)];
      tr.c_top += 1;
      tr.c[static_cast<size_t>(tr.c_top - 1)] = c;
      tr.dst = dst;
   }

   void generate(int idx, int entry, int exit) {   // nfa_node_m.F90:166-267
      if (idx == INVALID_INDEX) return;
      const TreeNode& n = tree.nodes[static_cast<size_t>(idx)];
      int entry_local = entry;
      switch (n.op) {
         case op_char:
            for (const Seg& s : n.c) add_transition(entry, exit, s);
            break;
         case op_empty: add_transition(entry, exit, SEG_EPSILON); break;
         case op_union:
            generate(n.left_i, entry, exit);
            generate(n.right_i, entry, exit);
            break;
         case op_closure: closure(idx, entry, exit); break;
         case op_concat: {
            int node1 = make_node();
            generate(n.left_i, entry, node1);
            generate(n.right_i, node1, exit);
            break;
         }
         case op_repeat: {
            int min_repeat = n.min_repeat, max_repeat = n.max_repeat;
            int num_1st = min_repeat - 1;
            if (max_repeat == INFINITE_REPEAT) num_1st += 1;
            for (int j = 1; j <= num_1st; ++j) {
               int node1 = make_node();
               generate(n.left_i, entry_local, node1);
               entry_local = node1;
            }
            int num_2nd = (min_repeat == 0) ? max_repeat - 1 : max_repeat - min_repeat;
            for (int j = 1; j <= num_2nd; ++j) {
               int node2 = make_node();
               generate(n.left_i, entry_local, node2);
               add_transition(node2, exit, SEG_EPSILON);
               entry_local = node2;
            }
            if (min_repeat == 0) add_transition(entry, exit, SEG_EPSILON);
            if (max_repeat == INFINITE_REPEAT) closure(idx, entry_local, exit);
            else generate(n.left_i, entry_local, exit);
            break;
         }
         default: throw LimitError{SYNTAX_ERR_THIS_SHOULD_NOT_HAPPEN};
      }
   }

   void closure(int idx, int entry, int exit) {   // nfa_node_m.F90:292-322
      int node1 = make_node();
      int node2 = make_node();
      add_transition(entry, node1, SEG_EPSILON);
      generate(tree.nodes[static_cast<size_t>(idx)].left_i, node1, node2);
      add_transition(node2, node1, SEG_EPSILON);
      add_transition(node1, exit, SEG_EPSILON);
   }
};
}   // namespace

Nfa build_nfa(const Tree& t, int max_states) {
   Nfa out;
   NfaBuilder b(t, max_states);
   try {
      out.entry = b.make_node();
      out.exit = b.make_node();
      b.generate(t.top, out.entry, out.exit);
   } catch (const LimitError& e) {
      out.status = e.code;
      return out;
   }
   // per-node sort + merge (nfa_node_m.F90:100-102, :667-692)
   for (int i = 1; i <= b.nfa_top; ++i)
      for (BuildTra& tr : b.g[static_cast<size_t>(i)].forward) {
         sort_segment_by_min(tr.c);
         merge_segments(tr.c);
         tr.c_top = static_cast<int>(tr.c.size());
      }
   // disjoin_nfa (nfa_node_m.F90:410-501)
   std::vector<Seg> seg_list;
   for (int i = 1; i <= b.nfa_top; ++i)
      for (const BuildTra& tr : b.g[static_cast<size_t>(i)].forward)
         if (tr.dst != NFA_NULL_TRANSITION)
            for (int k = 0; k < tr.c_top; ++k)
               if (tr.c[static_cast<size_t>(k)] != SEG_INIT) seg_list.push_back(tr.c[static_cast<size_t>(k)]);
   std::sort(seg_list.begin(), seg_list.end(),
             [](const Seg& a, const Seg& b2) { return a.min < b2.min || (a.min == b2.min && a.max < b2.max); });
   seg_list.erase(std::unique(seg_list.begin(), seg_list.end()), seg_list.end());
   disjoin(seg_list);
   out.all_segments = seg_list;
   out.nfa_top = b.nfa_top;
   out.nodes.resize(static_cast<size_t>(b.nfa_top) + 1);
   for (int i = 1; i <= b.nfa_top; ++i) {
      for (const BuildTra& tr : b.g[static_cast<size_t>(i)].forward) {
         // disjoin_nfa_each_transition (nfa_node_m.F90:508-554)
         std::vector<Seg> tmp;
         for (int k = 0; k < tr.c_top; ++k) {
            const Seg& seg = tr.c[static_cast<size_t>(k)];
            for (const Seg& piece : seg_list)
               if (seg.min <= piece.min && piece.max <= seg.max) tmp.push_back(piece);
         }
         NfaTransition nt;
         nt.c = tr.c;
         if (nt.c.size() < tmp.size()) nt.c.assign(tmp.size(), SEG_INIT);
         for (size_t k = 0; k < tmp.size(); ++k) nt.c[k] = tmp[k];
         nt.c_top = static_cast<int>(nt.c.size());
         nt.dst = tr.dst;
         out.nodes[static_cast<size_t>(i)].forward.push_back(nt);
      }
   }
   return out;
}

}   // namespace fxfe
