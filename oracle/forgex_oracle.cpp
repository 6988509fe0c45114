// TEST INFRASTRUCTURE ONLY -- never linked, imported or executed by the product (forgex_amd/).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
//
// CPU restatement of the reference's match path, statement by statement, with the reference's own
// algorithmic shape: per-call compile, NUL-wrapped text, restart-per-start search, and a lazy DFA
// whose every step re-runs an NFA move + epsilon closure.  Rows of SURVEY.md §8(a) covered here:
//   a1  operator__in            reference src/forgex.F90:74-160
//   a2  operator__match         reference src/forgex.F90:163-231
//   a3  subroutine__regex       reference src/forgex.F90:235-347
//   a4  do_matching_including   reference src/api_internal_m.F90:31-167
//   a5  do_matching_exactly     reference src/api_internal_m.F90:171-303
//   a6  construct/move/reachable reference src/automaton_m.F90:199-381
//   a7  epsilon closure         reference src/nfa/nfa_graph_m.F90:76-123, src/automaton_m.F90:121-151
//   a8  registered/register     reference src/lazy_dfa/lazy_dfa_graph_m.F90:122-143, src/automaton_m.F90:156-190
//   a9  UTF-8 stepping          reference src/essential/utf8_m.f90:168-191 (via the shared front end)
//   a10 symbol -> segment       reference src/essential/segment_m.F90:296-322
//   a11 get_index_list_forward  reference src/essential/utility_m.f90:58-117
// The pattern front end (tokenizer, parser, literal extraction, NFA build: rows a12/a14) is the
// product's own statement-level restatement in forgex_amd/csrc/frontend.cpp; the dependency points
// from this oracle to that file, never the other way.  PARITY PINNED: this oracle is checked against
// tests/golden/ref_tests.tsv (every assertion of the reference's test programs, recorded from the real
// reference) and against oracle/_ref/ref_driver (the real reference built with flang) by tests/.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../forgex_amd/csrc/frontend.hpp"

using namespace fxfe;

namespace {

constexpr int DFA_INVALID_INDEX = 0;
constexpr int ACCEPTED_EMPTY = -2;
constexpr int DFA_STATE_HARD_LIMIT = 1024 * 16 + 1;   // parameters_m.f90:118-122
constexpr int LIT_OPTS_INDEX_UNIT = 32;

using StateSet = std::vector<uint8_t>;   // logical vec(nfa_top), 1-based

struct Automaton {   // automaton_t, automaton_m.F90:28-49
   Nfa nfa;
   std::vector<StateSet> sets;      // DFA nodes, 1-based
   std::vector<uint8_t> accepted;   // 1-based
   std::map<StateSet, int> index;   // the reference's linear "registered" scan, as a map
   std::unordered_map<uint64_t, int> memo;   // (state, code point) -> destination; the reference recomputes every time
   int initial_index = -1;
   bool overflow = false;

   void mark_epsilon(StateSet& set, int idx) const {   // nfa_graph_m.F90:76-104 / automaton_m.F90:121-151
      set[static_cast<size_t>(idx)] = 1;
      for (const NfaTransition& tr : nfa.nodes[static_cast<size_t>(idx)].forward) {
         if (tr.c.empty()) continue;
         if (tr.is_epsilon() && tr.dst != NFA_NULL_TRANSITION && !(tr.dst != 0 && set[static_cast<size_t>(tr.dst)])) {
            mark_epsilon(set, tr.dst);
         }
      }
   }

   int register_state(const StateSet& set) {   // automaton_m.F90:156-190
      auto it = index.find(set);
      if (it != index.end()) return it->second;
      if (static_cast<int>(sets.size()) >= DFA_STATE_HARD_LIMIT) {
         overflow = true;
         return DFA_INVALID_INDEX;
      }
      sets.push_back(set);
      accepted.push_back(set[static_cast<size_t>(nfa.exit)]);
      int i = static_cast<int>(sets.size()) - 1;
      index.emplace(set, i);
      return i;
   }

   void init() {   // automaton_m.F90:66-100
      sets.assign(1, StateSet());
      accepted.assign(1, 0);
      StateSet closure(static_cast<size_t>(nfa.nfa_top) + 1, 0);
      mark_epsilon(closure, nfa.entry);
      initial_index = register_state(closure);
   }

   int construct(int curr, int32_t code) {   // automaton_m.F90:333-381 (+ :199-267 get_reachable)
      uint64_t key = (static_cast<uint64_t>(static_cast<uint32_t>(curr)) << 32) | static_cast<uint32_t>(code);
      auto m = memo.find(key);
      if (m != memo.end()) return m->second;
      const StateSet& cur = sets[static_cast<size_t>(curr)];
      StateSet next(static_cast<size_t>(nfa.nfa_top) + 1, 0);
      for (int i = 1; i <= nfa.nfa_top; ++i) {
         if (!cur[static_cast<size_t>(i)]) continue;
         for (const NfaTransition& tr : nfa.nodes[static_cast<size_t>(i)].forward)
            if (tr.dst != NFA_NULL_TRANSITION && tr.accepts(code)) next[static_cast<size_t>(tr.dst)] = 1;
      }
      // collect_epsilon_transition, nfa_graph_m.F90:107-123
      for (int i = 1; i <= nfa.nfa_top; ++i)
         if (next[static_cast<size_t>(i)]) mark_epsilon(next, i);
      bool any = false;
      for (int i = 1; i <= nfa.nfa_top; ++i) any = any || next[static_cast<size_t>(i)];
      int dst = any ? register_state(next) : DFA_INVALID_INDEX;
      memo.emplace(key, dst);
      return dst;
   }
};

int32_t symbol_code(const std::string& str, int ci, int next_ci, bool valid) {   // segment_m.F90:296-322
   if (!valid) return 65535;   // make_replacement_char, utf8_m.f90:433-438
   return ichar_utf8(str.substr(static_cast<size_t>(ci - 1), static_cast<size_t>(next_ci - ci)));
}

// utility_m.f90:58-117
bool get_index_list_forward(const std::string& text, const std::string& prefix, const std::string& suffix,
                            std::vector<int>& index_array) {
   int len_pre = static_cast<int>(prefix.size());
   if (len_pre == 0) return false;
   index_array.assign(LIT_OPTS_INDEX_UNIT, INVALID_CHAR_INDEX);
   int siz = LIT_OPTS_INDEX_UNIT;
   int idx = f_index(text, prefix);
   int suf_idx = f_index(text, suffix, true);
   if (suf_idx == 0) suf_idx = INVALID_CHAR_INDEX;
   if (idx <= 0) return true;
   else if (suf_idx != INVALID_CHAR_INDEX) {
      if (idx <= suf_idx) index_array[0] = idx;
   } else {
      index_array[0] = idx;
   }
   int offset = idx + len_pre - 1;
   int i = 2;
   int text_len = static_cast<int>(text.size());
   while (offset < text_len) {
      idx = f_index(text.substr(static_cast<size_t>(offset)), prefix);
      if (idx <= 0) break;
      index_array[static_cast<size_t>(i - 1)] = idx + offset;
      ++i;
      if (i > siz) {
         index_array.resize(static_cast<size_t>(siz) * 2, INVALID_CHAR_INDEX);
         siz *= 2;
      }
      offset = offset + idx + len_pre - 1;
      if (suf_idx != INVALID_CHAR_INDEX && offset > suf_idx) break;
   }
   return true;
}

// api_internal_m.F90:31-167
void do_matching_including(Automaton& a, const std::string& string, int& from, int& to, const std::string& prefix,
                           const std::string& suffix) {
   std::string str = std::string(1, '\0') + string + std::string(1, '\0');
   int len_str = static_cast<int>(str.size());
   from = 0;
   to = 0;
   bool do_brute_force = f_eq(prefix, "");
   int suf_idx = INVALID_CHAR_INDEX;
   int cur_i = a.initial_index;
   if (string.size() <= 1 && f_eq(string, "")) {
      if (a.accepted[static_cast<size_t>(cur_i)]) {
         from = ACCEPTED_EMPTY;
         to = ACCEPTED_EMPTY;
      }
      return;
   }
   std::vector<int> index_list;
   if (!do_brute_force) {
      if (!get_index_list_forward(str, prefix, suffix, index_list)) return;
      if (index_list[0] == INVALID_CHAR_INDEX) do_brute_force = true;
   }
   int i, start;
   if (do_brute_force) {
      i = 1;
      start = i;
   } else {
      if (index_list[0] == 2) {
         start = 1;
         i = 0;
      } else {
         i = 1;
         start = index_list[0];
      }
      if (!f_eq(suffix, "")) {
         suf_idx = f_index(string, suffix, true);
         if (suf_idx == 0) return;
      }
   }
   while (start < len_str) {
      int max_match = 0;
      int ci = start;
      cur_i = a.initial_index;
      if (suf_idx != INVALID_CHAR_INDEX && suf_idx < ci) break;
      while (cur_i != DFA_INVALID_INDEX) {
         if (a.accepted[static_cast<size_t>(cur_i)] && ci != start) max_match = ci;
         if (ci > len_str) break;
         int next_ci;
         bool valid;
         next_idxutf8_strict(str, ci, next_ci, valid);
         int dst_i = a.construct(cur_i, symbol_code(str, ci, next_ci, valid));
         cur_i = dst_i;
         ci = next_ci;
      }
      if (max_match > 0) {
         from = start - 1;
         if (from == 0) from = 1;
         if (max_match >= len_str) to = static_cast<int>(string.size());
         else to = max_match - 2;
         return;
      }
      if (do_brute_force) {
         bool valid;
         int nxt;
         next_idxutf8_strict(str, start, nxt, valid);
         start = nxt;
         continue;
      }
      ++i;
      if (i <= static_cast<int>(index_list.size())) {
         start = index_list[static_cast<size_t>(i - 1)];
         if (start == INVALID_CHAR_INDEX) return;
      } else {
         return;
      }
   }
}

// api_internal_m.F90:171-303
bool do_matching_exactly(Automaton& a, const std::string& string, const std::string& prefix, const std::string& suffix) {
   int len_pre = static_cast<int>(prefix.size()), len_suf = static_cast<int>(suffix.size());
   int n = static_cast<int>(string.size());
   bool matches_pre = true, matches_post = true;
   if (n > 0 && len_pre > 0)
      if (f_eq(prefix, string) && len_pre == n) return true;
   if (len_pre > n || len_suf > n) return false;
   bool empty_pre = f_eq(prefix, ""), empty_post = f_eq(suffix, "");
   if (n > 0) {
      if (!empty_pre) matches_pre = f_eq(string.substr(0, static_cast<size_t>(len_pre)), prefix);
      if (!empty_post) matches_post = f_eq(string.substr(static_cast<size_t>(n - len_suf)), suffix);
   } else {
      matches_pre = len_pre == 0;
      matches_post = len_suf == 0;
   }
   bool runs_engine = (empty_pre || matches_pre) && (empty_post || matches_post);
   if (!runs_engine) return false;
   int cur_i = a.initial_index;
   if (n == 0) return a.accepted[static_cast<size_t>(cur_i)] != 0;
   int max_match = 0;
   int ci = 1;
   std::string str = std::string(1, '\0') + string + std::string(1, '\0');
   int len_str = static_cast<int>(str.size());
   while (cur_i != DFA_INVALID_INDEX) {
      if (a.accepted[static_cast<size_t>(cur_i)]) max_match = ci;
      if (ci > len_str) break;
      int next_ci;
      bool valid;
      next_idxutf8_strict(str, ci, next_ci, valid);
      int dst_i = a.construct(cur_i, symbol_code(str, ci, next_ci, valid));
      if (dst_i == DFA_INVALID_INDEX && ci == 1) {
         ci = 2;
         next_idxutf8_strict(str, ci, next_ci, valid);
         dst_i = a.construct(cur_i, symbol_code(str, ci, next_ci, valid));
      }
      cur_i = dst_i;
      ci = next_ci;
   }
   return max_match >= n + 2;
}

bool is_there_caret_at_the_top(const std::string& pattern) {   // utility_m.f90:23-35
   std::string buff = f_adjustl(pattern);
   if (buff.empty()) return false;
   return buff[0] == '^';
}
bool is_there_dollar_at_the_end(const std::string& pattern) {   // utility_m.f90:40-53
   std::string buff = f_trim(pattern);
   if (buff.empty()) return false;
   return buff[buff.size() - 1] == '$';
}

struct Compiled {
   Tree tree;
   Literals lit;
   Automaton a;
   bool have_automaton = false;
};

constexpr int ORACLE_NFA_LIMIT = 1 << 20;

void ensure_automaton(Compiled& c) {
   if (c.have_automaton) return;
   c.a.nfa = build_nfa(c.tree, ORACLE_NFA_LIMIT);
   c.a.init();
   c.have_automaton = true;
}

// ---- public operations -------------------------------------------------------------------------------
struct RegexResult {
   int from = 0, to = 0, length = 0, status = 0;
   bool matched = false;          // a span was returned
   bool accepted_empty = false;   // ACCEPTED_EMPTY: `.in.` is true although regex() returns '' (forgex.F90:146-149, :323-329)
};

bool op_in(const std::string& pattern, const std::string& str) {   // forgex.F90:74-160
   Compiled c;
   c.tree.build(f_trim(pattern));
   if (!c.tree.is_valid) return false;
   c.lit = extract_literal(c.tree);
   if (!f_eq(c.lit.all, "")) {
      int from = f_index(str, c.lit.all);
      int to = INVALID_CHAR_INDEX;
      if (from > 0) to = from + static_cast<int>(c.lit.all.size()) - 1;
      return from > 0 && to > 0;
   }
   ensure_automaton(c);
   int from, to;
   do_matching_including(c.a, str, from, to, c.lit.prefix, c.lit.suffix);
   if (from == ACCEPTED_EMPTY && to == ACCEPTED_EMPTY) return true;
   return from > 0 && to > 0;
}

bool op_match(const std::string& pattern, const std::string& str) {   // forgex.F90:163-231
   std::string buff;
   if (is_there_caret_at_the_top(pattern)) buff = pattern.substr(1);
   else buff = pattern;
   if (is_there_dollar_at_the_end(pattern)) {
      int n = f_len_trim(pattern) - 1;
      if (n < 0) n = 0;
      buff = buff.substr(0, std::min(static_cast<size_t>(n), buff.size()));
   }
   Compiled c;
   c.tree.build(buff);
   if (!c.tree.is_valid) return false;
   c.lit = extract_literal(c.tree);
   if (!f_eq(c.lit.all, "")) {
      if (str.size() == c.lit.all.size()) return str == c.lit.all;
   }
   ensure_automaton(c);
   return do_matching_exactly(c.a, str, c.lit.prefix, c.lit.suffix);
}

RegexResult op_regex(const std::string& pattern, const std::string& text) {   // forgex.F90:235-347
   RegexResult r;
   Compiled c;
   c.tree.build(f_trim(pattern));
   if (!c.tree.is_valid) {
      r.from = r.to = INVALID_CHAR_INDEX;
      r.length = 0;
      r.status = c.tree.code;
      return r;
   }
   c.lit = extract_literal(c.tree);
   if (!f_eq(c.lit.all, "")) {
      int from_l = f_index(text, c.lit.all);
      int to_l = INVALID_CHAR_INDEX;
      if (from_l > 0) to_l = from_l + static_cast<int>(c.lit.all.size()) - 1;
      if (from_l > 0 && to_l > 0) {
         r.from = from_l;
         r.to = to_l;
         r.length = static_cast<int>(c.lit.all.size());
         r.matched = true;
      }
      return r;
   }
   ensure_automaton(c);
   int from_l, to_l;
   do_matching_including(c.a, text, from_l, to_l, c.lit.prefix, c.lit.suffix);
   if (from_l == ACCEPTED_EMPTY && to_l == ACCEPTED_EMPTY) {
      r.accepted_empty = true;
      return r;
   }
   if (from_l > 0 && to_l > 0) {
      r.from = from_l;
      r.to = to_l;
      r.length = to_l - from_l + 1;
      r.matched = true;
   }
   return r;
}

// ---- hex helpers / CLI ---------------------------------------------------------------------------------
std::string unhex(const std::string& h) {
   if (h == "-") return std::string();
   std::string s;
   for (size_t i = 0; i + 1 < h.size(); i += 2) s.push_back(static_cast<char>(std::stoi(h.substr(i, 2), nullptr, 16)));
   return s;
}
std::string tohex(const std::string& s) {
   if (s.empty()) return "-";
   static const char* d = "0123456789ABCDEF";
   std::string h;
   for (unsigned char ch : s) {
      h.push_back(d[ch >> 4]);
      h.push_back(d[ch & 15]);
   }
   return h;
}

}   // namespace

// ---- C ABI for ctypes (tests / bench cpu_baseline only) -----------------------------------------------------
#pragma GCC visibility push(default)   // built with -fvisibility=hidden: the checker exports its C entry points only
extern "C" {

int fxo_in(const char* pat, int64_t plen, const char* txt, int64_t tlen) {
   return op_in(std::string(pat, static_cast<size_t>(plen)), std::string(txt, static_cast<size_t>(tlen))) ? 1 : 0;
}
int fxo_match(const char* pat, int64_t plen, const char* txt, int64_t tlen) {
   return op_match(std::string(pat, static_cast<size_t>(plen)), std::string(txt, static_cast<size_t>(tlen))) ? 1 : 0;
}
int fxo_regex(const char* pat, int64_t plen, const char* txt, int64_t tlen, int32_t* from, int32_t* to, int32_t* length,
              int32_t* status) {
   RegexResult r = op_regex(std::string(pat, static_cast<size_t>(plen)), std::string(txt, static_cast<size_t>(tlen)));
   if (from) *from = r.from;
   if (to) *to = r.to;
   if (length) *length = r.length;
   if (status) *status = r.status;
   return r.matched ? 1 : 0;
}
int fxo_valid(const char* pat, int64_t plen) {
   Tree t;
   t.build(f_trim(std::string(pat, static_cast<size_t>(plen))));
   return t.is_valid ? 1 : 0;
}
// op: 0 = .in. (flags only), 1 = .match., 2 = regex (`.in.` verdict in flags + from/to; invalid pattern: -9999).  Every row pays the full per-call
// compile, exactly as the reference's elemental operators do (forgex.F90:98,139-140).
void fxo_batch(int op, const char* pat, int64_t plen, const uint8_t* rows, int64_t n, int64_t row_len, uint8_t* flags,
               int32_t* from, int32_t* to, int nthreads) {
   std::string p(pat, static_cast<size_t>(plen));
   (void)nthreads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
   for (int64_t i = 0; i < n; ++i) {
      std::string s(reinterpret_cast<const char*>(rows) + i * row_len, static_cast<size_t>(row_len));
      if (op == 0) {
         flags[i] = op_in(p, s) ? 1 : 0;
      } else if (op == 1) {
         flags[i] = op_match(p, s) ? 1 : 0;
      } else {
         RegexResult r = op_regex(p, s);
         flags[i] = (r.matched || r.accepted_empty) ? 1 : 0;   // the `.in.` verdict of the same call
         if (from) from[i] = r.from;
         if (to) to[i] = r.to;
      }
   }
}
}

#ifdef FXO_MAIN
// Same line protocol as oracle/ref_driver.f90 (I/M/R/V/L), so outputs can be diffed byte for byte.
int main() {
   std::string line;
   while (std::getline(std::cin, line)) {
      if (line.empty()) continue;
      std::vector<std::string> f;
      size_t p = 0;
      while (p < line.size()) {
         while (p < line.size() && line[p] == ' ') ++p;
         if (p >= line.size()) break;
         size_t q = line.find(' ', p);
         if (q == std::string::npos) q = line.size();
         f.push_back(line.substr(p, q - p));
         p = q;
      }
      if (f.size() < 3) {
         std::puts("E bad-line");
         continue;
      }
      std::string pat = unhex(f[1]), txt = unhex(f[2]);
      if (f[0] == "I") {
         std::printf("I %c\n", op_in(pat, txt) ? 'T' : 'F');
      } else if (f[0] == "M") {
         std::printf("M %c\n", op_match(pat, txt) ? 'T' : 'F');
      } else if (f[0] == "R") {
         RegexResult r = op_regex(pat, txt);
         std::string sub = r.matched ? txt.substr(static_cast<size_t>(r.from - 1), static_cast<size_t>(r.to - r.from + 1))
                                     : std::string();
         std::printf("R %d %d %d %d %s\n", r.from, r.to, r.length, r.status, tohex(sub).c_str());
      } else if (f[0] == "V") {
         Tree t;
         t.build(f_trim(pat));
         std::printf("V %c\n", t.is_valid ? 'T' : 'F');
      } else if (f[0] == "L") {
         Tree t;
         t.build(f_trim(pat));
         if (!t.is_valid) {
            std::puts("L F - - -");
         } else {
            Literals l = extract_literal(t);
            std::printf("L T %s %s %s\n", tohex(l.all).c_str(), tohex(l.prefix).c_str(), tohex(l.suffix).c_str());
         }
      } else {
         std::puts("E bad-op");
      }
      std::fflush(stdout);
   }
   return 0;
}
#endif
