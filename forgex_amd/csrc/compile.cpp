// See compile.hpp / program.h.
#include "compile.hpp"

#include <algorithm>
#include <cstring>
#include <map>

using namespace fxfe;

namespace fxc {
namespace {

using Bits = std::vector<uint64_t>;
inline bool bt(const Bits& b, int i) { return (b[static_cast<size_t>(i) >> 6] >> (i & 63)) & 1u; }
inline void bs(Bits& b, int i) { b[static_cast<size_t>(i) >> 6] |= (uint64_t(1) << (i & 63)); }
inline void bor(Bits& a, const Bits& b) {
   for (size_t i = 0; i < a.size(); ++i) a[i] |= b[i];
}
inline bool bany(const Bits& b) {
   for (uint64_t w : b)
      if (w) return true;
   return false;
}

struct FlatTra {
   int src, dst;
   std::vector<int> acc;   // interval ids whose code points this transition consumes
};

struct Dfa {
   int n = 0;                 // number of states
   int ncol = 0;              // number of columns (intervals, later classes)
   std::vector<int> T;        // [n][ncol]
   std::vector<uint8_t> out;  // accept (A) / hit (R)
   int init = 0;
   bool ok = true;
};

// Moore partition refinement; keeps `keep0` (the dead state of A) as state 0 when >= 0.
void minimise(Dfa& d, int keep0) {
   int n = d.n, nc = d.ncol;
   std::vector<int> block(static_cast<size_t>(n));
   for (int s = 0; s < n; ++s) block[static_cast<size_t>(s)] = d.out[static_cast<size_t>(s)];
   int nblocks = 0;
   {
      std::map<int, int> m;
      for (int s = 0; s < n; ++s) {
         auto it = m.find(block[static_cast<size_t>(s)]);
         if (it == m.end()) it = m.emplace(block[static_cast<size_t>(s)], static_cast<int>(m.size())).first;
         block[static_cast<size_t>(s)] = it->second;
      }
      nblocks = static_cast<int>(m.size());
   }
   while (true) {
      std::map<std::vector<int>, int> m;
      std::vector<int> nb(static_cast<size_t>(n));
      std::vector<int> sig(static_cast<size_t>(nc) + 1);
      for (int s = 0; s < n; ++s) {
         sig[0] = block[static_cast<size_t>(s)];
         for (int c = 0; c < nc; ++c) sig[static_cast<size_t>(c) + 1] = block[static_cast<size_t>(d.T[static_cast<size_t>(s) * nc + c])];
         auto it = m.find(sig);
         if (it == m.end()) it = m.emplace(sig, static_cast<int>(m.size())).first;
         nb[static_cast<size_t>(s)] = it->second;
      }
      int k = static_cast<int>(m.size());
      block.swap(nb);
      if (k == nblocks) break;
      nblocks = k;
   }
   // renumber: dead block first (if any), then non-output blocks, then output blocks -- only REACHABLE blocks kept
   std::vector<int> rep(static_cast<size_t>(nblocks), -1);
   for (int s = 0; s < n; ++s)
      if (rep[static_cast<size_t>(block[static_cast<size_t>(s)])] < 0) rep[static_cast<size_t>(block[static_cast<size_t>(s)])] = s;
   std::vector<uint8_t> reach(static_cast<size_t>(nblocks), 0);
   std::vector<int> stack{block[static_cast<size_t>(d.init)]};
   reach[static_cast<size_t>(stack[0])] = 1;
   if (keep0 >= 0) reach[static_cast<size_t>(block[static_cast<size_t>(keep0)])] = 1;
   while (!stack.empty()) {
      int b = stack.back();
      stack.pop_back();
      int s = rep[static_cast<size_t>(b)];
      for (int c = 0; c < nc; ++c) {
         int nb2 = block[static_cast<size_t>(d.T[static_cast<size_t>(s) * nc + c])];
         if (!reach[static_cast<size_t>(nb2)]) {
            reach[static_cast<size_t>(nb2)] = 1;
            stack.push_back(nb2);
         }
      }
   }
   std::vector<int> newid(static_cast<size_t>(nblocks), -1);
   int next = 0;
   if (keep0 >= 0) newid[static_cast<size_t>(block[static_cast<size_t>(keep0)])] = next++;
   for (int pass = 0; pass < 2; ++pass)
      for (int b = 0; b < nblocks; ++b)
         if (reach[static_cast<size_t>(b)] && newid[static_cast<size_t>(b)] < 0 && d.out[static_cast<size_t>(rep[static_cast<size_t>(b)])] == pass)
            newid[static_cast<size_t>(b)] = next++;
   Dfa r;
   r.n = next;
   r.ncol = nc;
   r.T.assign(static_cast<size_t>(next) * nc, 0);
   r.out.assign(static_cast<size_t>(next), 0);
   for (int b = 0; b < nblocks; ++b) {
      int id = newid[static_cast<size_t>(b)];
      if (id < 0) continue;
      int s = rep[static_cast<size_t>(b)];
      r.out[static_cast<size_t>(id)] = d.out[static_cast<size_t>(s)];
      for (int c = 0; c < nc; ++c) r.T[static_cast<size_t>(id) * nc + c] = newid[static_cast<size_t>(block[static_cast<size_t>(d.T[static_cast<size_t>(s) * nc + c])])];
   }
   r.init = newid[static_cast<size_t>(block[static_cast<size_t>(d.init)])];
   d = r;
}

struct Blob {
   std::vector<uint8_t> b;
   uint32_t put(const void* p, size_t n) {
      while (b.size() % 16) b.push_back(0);
      uint32_t off = static_cast<uint32_t>(b.size());
      const uint8_t* q = static_cast<const uint8_t*>(p);
      if (n) b.insert(b.end(), q, q + n);
      if (n == 0) b.push_back(0);
      return off;
   }
};

Program finish(FxpHeader h, Blob& bl) {
   while (bl.b.size() % 16) bl.b.push_back(0);
   h.magic = FXP_MAGIC;
   h.version = FXP_VERSION;
   h.total_bytes = static_cast<uint32_t>(bl.b.size());
   std::memcpy(bl.b.data(), &h, sizeof(h));
   Program p;
   p.status = static_cast<int>(h.status);
   p.blob.swap(bl.b);
   return p;
}

Program invalid_program(int status) {
   Blob bl;
   bl.b.assign(sizeof(FxpHeader), 0);
   FxpHeader h;
   std::memset(&h, 0, sizeof(h));
   h.mode = FXP_MODE_INVALID;
   h.status = static_cast<uint32_t>(status);
   return finish(h, bl);
}

std::vector<int32_t> decode_chars(const std::string& s) {
   std::vector<int32_t> out;
   int i = 1, n = static_cast<int>(s.size());
   while (i <= n) {
      int nxt;
      bool valid;
      next_idxutf8_strict(s, i, nxt, valid);
      out.push_back(valid ? ichar_utf8(s.substr(static_cast<size_t>(i - 1), static_cast<size_t>(nxt - i))) : 65535);
      i = nxt;
   }
   return out;
}

bool border_free(const std::string& p) {
   for (size_t k = 1; k < p.size(); ++k)
      if (p.compare(0, k, p, p.size() - k, k) == 0) return false;
   return true;
}

// interval starts, class of each interval, ASCII class table and the two-level BMP page map
void emit_class_map(Blob& bl, FxpHeader& h, const std::vector<int32_t>& bounds, int nI, const std::vector<int>& cls_of) {
   auto interval_of = [&](int32_t code) {
      return static_cast<int>(std::upper_bound(bounds.begin(), bounds.end(), code) - bounds.begin()) - 1;
   };
   h.n_bounds = static_cast<uint32_t>(nI);
   std::vector<int32_t> b32(bounds.begin(), bounds.begin() + nI);
   h.off_bounds = bl.put(b32.data(), b32.size() * 4);
   std::vector<uint16_t> bc(static_cast<size_t>(nI));
   for (int k = 0; k < nI; ++k) bc[static_cast<size_t>(k)] = static_cast<uint16_t>(cls_of[static_cast<size_t>(k)]);
   h.off_bound_cls = bl.put(bc.data(), bc.size() * 2);
   std::vector<uint16_t> ac(128);
   for (int c = 0; c < 128; ++c) ac[static_cast<size_t>(c)] = static_cast<uint16_t>(cls_of[static_cast<size_t>(interval_of(c))]);
   h.off_ascii_cls = bl.put(ac.data(), ac.size() * 2);
   h.cls_nul = static_cast<uint32_t>(cls_of[static_cast<size_t>(interval_of(0))]);
   h.cls_ffff = static_cast<uint32_t>(cls_of[static_cast<size_t>(interval_of(65535))]);
   // two-level class map of the BMP: the on-device decoder classifies a 2-/3-byte character with two table reads
   // instead of a binary search over the interval starts
   std::vector<uint16_t> page_of(1024);
   std::vector<uint16_t> pages;
   std::map<std::vector<uint16_t>, uint16_t> seen;
   for (int pg = 0; pg < 1024; ++pg) {
      std::vector<uint16_t> v(64);
      int iv = interval_of(pg * 64);
      for (int k = 0; k < 64; ++k) {
         int32_t code = pg * 64 + k;
         while (iv + 1 < nI && bounds[static_cast<size_t>(iv) + 1] <= code) ++iv;
         v[static_cast<size_t>(k)] = static_cast<uint16_t>(cls_of[static_cast<size_t>(iv)]);
      }
      auto it = seen.find(v);
      if (it == seen.end()) {
         it = seen.emplace(v, static_cast<uint16_t>(seen.size())).first;
         pages.insert(pages.end(), v.begin(), v.end());
      }
      page_of[static_cast<size_t>(pg)] = it->second;
   }
   h.n_pages = static_cast<uint32_t>(seen.size());
   h.off_cls_page = bl.put(page_of.data(), page_of.size() * 2);
   h.off_cls_pages = bl.put(pages.data(), pages.size() * 2);
}

std::vector<uint32_t> to_words(const Bits& b, uint32_t words) {
   std::vector<uint32_t> w(words, 0);
   for (uint32_t i = 0; i < words; ++i) {
      const uint64_t q = (i >> 1) < b.size() ? b[i >> 1] : 0;
      w[i] = static_cast<uint32_t>(q >> ((i & 1) * 32));
   }
   return w;
}

// The DFA does not fit: emit the NFA itself for on-device simulation of state SETS (same algorithm as the DFA path --
// reverse unanchored pass for the leftmost start, forward anchored pass for the longest end -- with bitsets as states).
Program compile_nfa_sim(const Nfa& nfa, const Literals& lit, int op, const std::vector<int32_t>& bounds, int nI,
                        const std::vector<FlatTra>& tras, const std::vector<Bits>& clos, const std::vector<Bits>& rclos) {
   const int N = nfa.nfa_top;
   const uint32_t words = static_cast<uint32_t>(N + 32) / 32;
   // classes: intervals consumed by exactly the same transitions
   std::vector<std::vector<int>> sig(static_cast<size_t>(nI));
   for (size_t t = 0; t < tras.size(); ++t)
      for (int k : tras[t].acc) sig[static_cast<size_t>(k)].push_back(static_cast<int>(t));
   std::vector<int> cls_of(static_cast<size_t>(nI));
   std::map<std::vector<int>, int> m;
   for (int k = 0; k < nI; ++k) {
      auto it = m.find(sig[static_cast<size_t>(k)]);
      if (it == m.end()) it = m.emplace(sig[static_cast<size_t>(k)], static_cast<int>(m.size())).first;
      cls_of[static_cast<size_t>(k)] = it->second;
   }
   const int ncls = static_cast<int>(m.size());
   const size_t table_words = static_cast<size_t>(ncls) * (N + 1) * words;
   if (table_words * 8 > (size_t(1) << 30)) return invalid_program(FX_ERR_NFA_LIMIT);   // 2 tables x 4 bytes per word: keep under 1 GiB

   Blob bl;
   bl.b.assign(sizeof(FxpHeader), 0);
   FxpHeader h;
   std::memset(&h, 0, sizeof(h));
   h.mode = op == OP_SEARCH ? FXP_MODE_SEARCH_ENGINE : FXP_MODE_MATCH_ENGINE;
   h.flags = FXP_F_NFA_SIM | FXP_F_HAS_R;
   h.n_classes = static_cast<uint32_t>(ncls);
   h.len_prefix = static_cast<uint32_t>(lit.prefix.size());
   h.len_suffix = static_cast<uint32_t>(lit.suffix.size());
   h.len_all = static_cast<uint32_t>(lit.all.size());
   if (bt(clos[static_cast<size_t>(nfa.entry)], nfa.exit)) h.flags |= FXP_F_INIT_ACCEPTING;
   if (!f_eq(lit.prefix, "")) h.flags |= FXP_F_PREFILTER;
   if (!f_eq(lit.suffix, "")) h.flags |= FXP_F_HAS_SUFFIX;
   if (op == OP_MATCH && !f_eq(lit.all, "")) h.flags |= FXP_F_MATCH_LITERAL;
   emit_class_map(bl, h, bounds, nI, cls_of);
   h.off_prefix = bl.put(lit.prefix.data(), lit.prefix.size());
   h.off_suffix = bl.put(lit.suffix.data(), lit.suffix.size());
   h.off_all = bl.put(lit.all.data(), lit.all.size());
   h.nfa_N = static_cast<uint32_t>(N);
   h.nfa_words = words;
   h.nfa_entry = static_cast<uint32_t>(nfa.entry);
   h.nfa_exit = static_cast<uint32_t>(nfa.exit);
   // per-class transition bitsets
   std::vector<uint32_t> fwdT(table_words, 0), revT(table_words, 0);
   std::vector<std::vector<uint8_t>> acc_cls(tras.size(), std::vector<uint8_t>(static_cast<size_t>(ncls), 0));
   for (size_t t = 0; t < tras.size(); ++t)
      for (int k : tras[t].acc) acc_cls[t][static_cast<size_t>(cls_of[static_cast<size_t>(k)])] = 1;
   for (size_t t = 0; t < tras.size(); ++t) {
      const std::vector<uint32_t> cw = to_words(clos[static_cast<size_t>(tras[t].dst)], words);
      const std::vector<uint32_t> rw = to_words(rclos[static_cast<size_t>(tras[t].src)], words);
      for (int c = 0; c < ncls; ++c) {
         if (!acc_cls[t][static_cast<size_t>(c)]) continue;
         uint32_t* f = &fwdT[(static_cast<size_t>(c) * (N + 1) + tras[t].src) * words];
         uint32_t* r = &revT[(static_cast<size_t>(c) * (N + 1) + tras[t].dst) * words];
         for (uint32_t i = 0; i < words; ++i) {
            f[i] |= cw[i];
            r[i] |= rw[i];
         }
      }
   }
   const std::vector<uint32_t> init = to_words(clos[static_cast<size_t>(nfa.entry)], words);
   const std::vector<uint32_t> f0 = to_words(rclos[static_cast<size_t>(nfa.exit)], words);
   std::vector<uint32_t> rstart = f0;   // F0 + pre(NUL, F0), hit cleared: the trailing NUL is never a start
   for (int z = 1; z <= N; ++z)
      if ((f0[static_cast<size_t>(z) >> 5] >> (z & 31)) & 1u) {
         const uint32_t* r = &revT[(static_cast<size_t>(h.cls_nul) * (N + 1) + z) * words];
         for (uint32_t i = 0; i < words; ++i) rstart[i] |= r[i];
      }
   h.off_nfa_init = bl.put(init.data(), init.size() * 4);
   h.off_nfa_f0 = bl.put(f0.data(), f0.size() * 4);
   h.off_nfa_rstart = bl.put(rstart.data(), rstart.size() * 4);
   h.off_nfa_fwd = bl.put(fwdT.data(), fwdT.size() * 4);
   h.off_nfa_rev = bl.put(revT.data(), revT.size() * 4);
   // empty placeholders so that every offset stays inside the blob
   const uint32_t none = bl.put(nullptr, 0);
   h.off_TA = h.off_TR = h.off_accA = h.off_hitR = h.off_finalM = h.off_fastA = h.off_fastR = none;
   h.off_chain_cls = h.off_chain_TR = h.off_chain_TA = none;
   return finish(h, bl);
}

}   // namespace

Program compile_from_nfa(const Nfa& nfa, const Literals& lit, int op, const Limits& lim) {
   if (nfa.status != SYNTAX_VALID) return invalid_program(nfa.status);
   const int N = nfa.nfa_top;
   if (N > lim.max_nfa_states) return invalid_program(FX_ERR_NFA_LIMIT);
   const size_t W = (static_cast<size_t>(N) + 64) / 64;

   // ---- 1. code-point intervals induced by every segment edge ---------------------------------------
   std::vector<int32_t> bounds{0, 0x200000};
   for (int i = 1; i <= N; ++i)
      for (const NfaTransition& tr : nfa.nodes[static_cast<size_t>(i)].forward)
         for (const Seg& s : tr.c) {
            if (s.max < 0 || s.min > s.max || s.min >= 0x200000) continue;
            bounds.push_back(std::max<int32_t>(s.min, 0));
            bounds.push_back(std::min<int32_t>(s.max, 0x1FFFFF) + 1);
         }
   std::sort(bounds.begin(), bounds.end());
   bounds.erase(std::unique(bounds.begin(), bounds.end()), bounds.end());
   const int nI = static_cast<int>(bounds.size()) - 1;
   auto interval_of = [&](int32_t code) {
      return static_cast<int>(std::upper_bound(bounds.begin(), bounds.end(), code) - bounds.begin()) - 1;
   };

   // ---- 2. flatten transitions, epsilon closures -----------------------------------------------------
   std::vector<FlatTra> tras;
   std::vector<std::vector<int>> fwd(static_cast<size_t>(N) + 1), inc(static_cast<size_t>(N) + 1), eps(static_cast<size_t>(N) + 1);
   for (int i = 1; i <= N; ++i)
      for (const NfaTransition& tr : nfa.nodes[static_cast<size_t>(i)].forward) {
         if (tr.dst == NFA_NULL_TRANSITION || tr.dst < 1 || tr.dst > N) continue;
         if (tr.is_epsilon()) eps[static_cast<size_t>(i)].push_back(tr.dst);
         FlatTra ft;
         ft.src = i;
         ft.dst = tr.dst;
         for (int k = 0; k < nI; ++k)
            if (tr.accepts(bounds[static_cast<size_t>(k)])) ft.acc.push_back(k);
         if (!ft.acc.empty()) {
            fwd[static_cast<size_t>(i)].push_back(static_cast<int>(tras.size()));
            inc[static_cast<size_t>(tr.dst)].push_back(static_cast<int>(tras.size()));
            tras.push_back(std::move(ft));
         }
      }
   std::vector<Bits> clos(static_cast<size_t>(N) + 1, Bits(W, 0)), rclos(static_cast<size_t>(N) + 1, Bits(W, 0));
   for (int z = 1; z <= N; ++z) {
      std::vector<int> st{z};
      bs(clos[static_cast<size_t>(z)], z);
      while (!st.empty()) {
         int x = st.back();
         st.pop_back();
         for (int y : eps[static_cast<size_t>(x)])
            if (!bt(clos[static_cast<size_t>(z)], y)) {
               bs(clos[static_cast<size_t>(z)], y);
               st.push_back(y);
            }
      }
      for (int x = 1; x <= N; ++x)
         if (bt(clos[static_cast<size_t>(z)], x)) bs(rclos[static_cast<size_t>(x)], z);
   }

   // ---- 3. forward anchored DFA A ----------------------------------------------------------------------
   Dfa A;
   A.ncol = nI;
   {
      std::map<Bits, int> ids;
      std::vector<Bits> sets;
      sets.emplace_back(W, 0);   // state 0 = dead = empty set (DFA_INVALID_INDEX)
      ids.emplace(sets[0], 0);
      sets.push_back(clos[static_cast<size_t>(nfa.entry)]);
      ids.emplace(sets[1], 1);
      A.init = 1;
      std::vector<Bits> next(static_cast<size_t>(nI), Bits(W, 0));
      for (size_t s = 0; s < sets.size(); ++s) {
         for (auto& nb : next) std::fill(nb.begin(), nb.end(), 0);
         const Bits cur = sets[s];
         for (int x = 1; x <= N; ++x) {
            if (!bt(cur, x)) continue;
            for (int ti : fwd[static_cast<size_t>(x)]) {
               const FlatTra& ft = tras[static_cast<size_t>(ti)];
               for (int k : ft.acc) bor(next[static_cast<size_t>(k)], clos[static_cast<size_t>(ft.dst)]);
            }
         }
         A.T.resize((s + 1) * static_cast<size_t>(nI));
         for (int k = 0; k < nI; ++k) {
            auto it = ids.find(next[static_cast<size_t>(k)]);
            if (it == ids.end()) {
               if (static_cast<int>(sets.size()) >= lim.max_dfa_states) return compile_nfa_sim(nfa, lit, op, bounds, nI, tras, clos, rclos);
               it = ids.emplace(next[static_cast<size_t>(k)], static_cast<int>(sets.size())).first;
               sets.push_back(next[static_cast<size_t>(k)]);
            }
            A.T[s * static_cast<size_t>(nI) + static_cast<size_t>(k)] = it->second;
         }
      }
      A.n = static_cast<int>(sets.size());
      A.out.assign(static_cast<size_t>(A.n), 0);
      for (int s = 0; s < A.n; ++s) A.out[static_cast<size_t>(s)] = bt(sets[static_cast<size_t>(s)], nfa.exit) ? 1 : 0;
   }
   minimise(A, 0);

   // ---- 4. reverse unanchored DFA R (search only) ---------------------------------------------------------
   Dfa R;
   R.ncol = nI;
   R.ok = false;
   bool r_has_skip = false;   // R distinguishes the SKIP symbol (fast path may translate UTF-8 in place)
   int R_start_raw = 0;
   const int i_nul = interval_of(0);
   if (op == OP_SEARCH) {
      R.ok = true;
      const Bits& F0 = rclos[static_cast<size_t>(nfa.exit)];
      std::map<std::pair<Bits, int>, int> ids;
      std::vector<std::pair<Bits, int>> sets;
      sets.emplace_back(F0, 0);
      ids.emplace(sets[0], 0);
      R.init = 0;
      std::vector<Bits> ac(static_cast<size_t>(nI), Bits(W, 0));
      for (size_t s = 0; s < sets.size() && R.ok; ++s) {
         for (auto& b : ac) std::fill(b.begin(), b.end(), 0);
         const Bits cur = sets[s].first;
         for (int z = 1; z <= N; ++z) {
            if (!bt(cur, z)) continue;
            for (int ti : inc[static_cast<size_t>(z)]) {
               const FlatTra& ft = tras[static_cast<size_t>(ti)];
               for (int k : ft.acc) bs(ac[static_cast<size_t>(k)], ft.src);
            }
         }
         R.T.resize((s + 1) * static_cast<size_t>(nI));
         for (int k = 0; k < nI; ++k) {
            Bits pre(W, 0);
            for (int x = 1; x <= N; ++x)
               if (bt(ac[static_cast<size_t>(k)], x)) bor(pre, rclos[static_cast<size_t>(x)]);
            int hit = bt(pre, nfa.entry) ? 1 : 0;
            bor(pre, F0);
            std::pair<Bits, int> key(pre, hit);
            auto it = ids.find(key);
            if (it == ids.end()) {
               if (static_cast<int>(sets.size()) >= lim.max_dfa_states) {
                  R.ok = false;
                  break;
               }
               it = ids.emplace(key, static_cast<int>(sets.size())).first;
               sets.push_back(key);
            }
            R.T[s * static_cast<size_t>(nI) + static_cast<size_t>(k)] = it->second;
         }
      }
      if (R.ok) {
         R.n = static_cast<int>(sets.size());
         R.out.assign(static_cast<size_t>(R.n), 0);
         for (int s = 0; s < R.n; ++s) R.out[static_cast<size_t>(s)] = static_cast<uint8_t>(sets[static_cast<size_t>(s)].second);
         // start of the scan = state after the trailing NUL with its hit cleared (the trailing NUL is never a start:
         // `do while (start < len(str))`, api_internal_m.F90:108).  Add that state explicitly before minimising.
         int t = R.T[static_cast<size_t>(R.init) * nI + static_cast<size_t>(i_nul)];
         std::pair<Bits, int> key(sets[static_cast<size_t>(t)].first, 0);
         auto it = ids.find(key);
         if (it != ids.end()) {
            R_start_raw = it->second;
         } else {
            // same W, hit = 0: transitions identical to state t (they depend on W only)
            R_start_raw = R.n;
            R.n += 1;
            R.out.push_back(0);
            R.T.resize(static_cast<size_t>(R.n) * nI);
            for (int k = 0; k < nI; ++k) R.T[static_cast<size_t>(R_start_raw) * nI + k] = R.T[static_cast<size_t>(t) * nI + k];
         }
         R.init = R_start_raw;
         // Extra column SKIP (index nI): the symbol the fast kernel substitutes for continuation bytes inside a valid
         // multi-byte character.  It must leave W unchanged and CLEAR the hit (a hit belongs to the character's first
         // byte only), i.e. (W,1) -> (W,0), (W,0) -> itself.  The twins (W,0) are added where missing.
         const int nC = nI + 1;
         {
            std::vector<int> T2(static_cast<size_t>(R.n) * nC);
            for (int st = 0; st < R.n; ++st)
               for (int k = 0; k < nI; ++k) T2[static_cast<size_t>(st) * nC + k] = R.T[static_cast<size_t>(st) * nI + k];
            std::map<std::vector<int>, int> twin_of_row;   // states with hit = 0, keyed by their transition row (same W <=> same row
                                                           // is not guaranteed, so twins are created per hit state and merged by minimise)
            int n0 = R.n;
            for (int st = 0; st < n0; ++st) {
               if (!R.out[static_cast<size_t>(st)]) {
                  T2[static_cast<size_t>(st) * nC + nI] = st;
                  continue;
               }
               int tw = R.n++;
               R.out.push_back(0);
               T2.resize(static_cast<size_t>(R.n) * nC);
               for (int k = 0; k < nI; ++k) T2[static_cast<size_t>(tw) * nC + k] = T2[static_cast<size_t>(st) * nC + k];
               T2[static_cast<size_t>(tw) * nC + nI] = tw;
               T2[static_cast<size_t>(st) * nC + nI] = tw;
            }
            R.T.swap(T2);
            R.ncol = nC;
         }
         Dfa Ru = R;
         minimise(Ru, -1);
         if (Ru.n <= 8) {
            R = Ru;
            r_has_skip = true;
         } else {
            // more than 8 states once SKIP must be told apart: if the SKIP-blind automaton still fits the v_perm tables keep
            // that ASCII-only fast path, otherwise keep SKIP (class-indexed chain tables have no 8-state limit)
            Dfa Rn = R;
            for (int st = 0; st < Rn.n; ++st) Rn.T[static_cast<size_t>(st) * nC + nI] = st;
            minimise(Rn, -1);
            if (Rn.n <= 8) {
               R = Rn;
            } else {
               R = Ru;
               r_has_skip = true;
            }
         }
      }
   }

   // ---- 5. merge intervals with identical columns into classes -----------------------------------------------
   std::vector<int> cls_of(static_cast<size_t>(nI));
   int ncls = 0;
   {
      std::map<std::vector<int>, int> m;
      for (int k = 0; k < nI; ++k) {
         std::vector<int> col;
         for (int s = 0; s < A.n; ++s) col.push_back(A.T[static_cast<size_t>(s) * nI + k]);
         if (R.ok)
            for (int s = 0; s < R.n; ++s) col.push_back(R.T[static_cast<size_t>(s) * R.ncol + k]);
         auto it = m.find(col);
         if (it == m.end()) it = m.emplace(col, static_cast<int>(m.size())).first;
         cls_of[static_cast<size_t>(k)] = it->second;
      }
      ncls = static_cast<int>(m.size());
   }
   std::vector<int> rep_interval(static_cast<size_t>(ncls), -1);
   for (int k = 0; k < nI; ++k)
      if (rep_interval[static_cast<size_t>(cls_of[static_cast<size_t>(k)])] < 0) rep_interval[static_cast<size_t>(cls_of[static_cast<size_t>(k)])] = k;
   auto TA = [&](int s, int c) { return A.T[static_cast<size_t>(s) * nI + rep_interval[static_cast<size_t>(c)]]; };
   auto TR = [&](int s, int c) { return R.T[static_cast<size_t>(s) * R.ncol + rep_interval[static_cast<size_t>(c)]]; };
   auto class_of_code = [&](int32_t code) { return cls_of[static_cast<size_t>(interval_of(code))]; };

   // ---- 6. emit ---------------------------------------------------------------------------------------------------
   Blob bl;
   bl.b.assign(sizeof(FxpHeader), 0);
   FxpHeader h;
   std::memset(&h, 0, sizeof(h));
   h.status = 0;
   h.mode = op == OP_SEARCH ? FXP_MODE_SEARCH_ENGINE : FXP_MODE_MATCH_ENGINE;
   h.n_classes = static_cast<uint32_t>(ncls);
   h.n_bounds = static_cast<uint32_t>(nI);
   h.nA = static_cast<uint32_t>(A.n);
   h.nR = R.ok ? static_cast<uint32_t>(R.n) : 0;
   h.A_init = static_cast<uint32_t>(A.init);
   h.R_start = R.ok ? static_cast<uint32_t>(R.init) : 0;
   h.cls_nul = static_cast<uint32_t>(class_of_code(0));
   h.cls_ffff = static_cast<uint32_t>(class_of_code(65535));
   {
      int t = TA(A.init, static_cast<int>(h.cls_nul));
      h.M_start = static_cast<uint32_t>(t != 0 ? t : A.init);   // api_internal_m.F90:280-289
   }
   h.len_prefix = static_cast<uint32_t>(lit.prefix.size());
   h.len_suffix = static_cast<uint32_t>(lit.suffix.size());
   h.len_all = static_cast<uint32_t>(lit.all.size());
   if (A.out[static_cast<size_t>(A.init)]) h.flags |= FXP_F_INIT_ACCEPTING;
   const bool prefilter = !f_eq(lit.prefix, "");
   const bool has_suffix = !f_eq(lit.suffix, "");
   if (prefilter) h.flags |= FXP_F_PREFILTER;
   if (has_suffix) h.flags |= FXP_F_HAS_SUFFIX;
   if (R.ok) h.flags |= FXP_F_HAS_R;
   if (op == OP_MATCH && !f_eq(lit.all, "")) h.flags |= FXP_F_MATCH_LITERAL;

   emit_class_map(bl, h, bounds, nI, cls_of);
   std::vector<uint16_t> ac(128);
   for (int c = 0; c < 128; ++c) ac[static_cast<size_t>(c)] = static_cast<uint16_t>(class_of_code(c));
   std::vector<uint16_t> ta(static_cast<size_t>(A.n) * ncls);
   for (int s = 0; s < A.n; ++s)
      for (int c = 0; c < ncls; ++c) {
         int t = TA(s, c);
         ta[static_cast<size_t>(s) * ncls + c] = static_cast<uint16_t>(t | (A.out[static_cast<size_t>(t)] ? FXP_FLAG_BIT : 0));
      }
   h.off_TA = bl.put(ta.data(), ta.size() * 2);
   std::vector<uint16_t> tr;
   if (R.ok) {
      tr.resize(static_cast<size_t>(R.n) * ncls);
      for (int s = 0; s < R.n; ++s)
         for (int c = 0; c < ncls; ++c) {
            int t = TR(s, c);
            tr[static_cast<size_t>(s) * ncls + c] = static_cast<uint16_t>(t | (R.out[static_cast<size_t>(t)] ? FXP_FLAG_BIT : 0));
         }
   }
   h.off_TR = bl.put(tr.data(), tr.size() * 2);
   h.off_accA = bl.put(A.out.data(), A.out.size());
   h.off_hitR = bl.put(R.out.data(), R.ok ? R.out.size() : 0);
   std::vector<uint8_t> fin(static_cast<size_t>(A.n));
   for (int s = 0; s < A.n; ++s)   // api_internal_m.F90:258-302: accept at ci = n+2, or after the trailing NUL at n+3
      fin[static_cast<size_t>(s)] = (s != 0 && (A.out[static_cast<size_t>(s)] || A.out[static_cast<size_t>(TA(s, static_cast<int>(h.cls_nul)))])) ? 1 : 0;
   h.off_finalM = bl.put(fin.data(), fin.size());
   h.off_prefix = bl.put(lit.prefix.data(), lit.prefix.size());
   h.off_suffix = bl.put(lit.suffix.data(), lit.suffix.size());
   h.off_all = bl.put(lit.all.data(), lit.all.size());

   // ---- 7. fast path: <= 8 states per automaton, fused ASCII byte tables -----------------------------------------
   // Candidate-list search == brute-force search on pure-ASCII rows iff the prefix is a NECESSARY, non-self-overlapping
   // beginning of every non-empty match; the suffix is only consulted by the candidate-list driver.
   bool brute_equiv = op == OP_SEARCH && R.ok && !(prefilter && has_suffix);
   if (brute_equiv && prefilter) {
      bool ok = border_free(lit.prefix) && lit.prefix.find('\0') == std::string::npos;
      int q = A.init;
      std::vector<int32_t> codes = decode_chars(lit.prefix);
      for (size_t i = 0; ok && i < codes.size(); ++i) {
         if (i > 0 && A.out[static_cast<size_t>(q)]) ok = false;
         int iv = interval_of(codes[i]);
         int c = cls_of[static_cast<size_t>(iv)];
         if (!(bounds[static_cast<size_t>(iv)] == codes[i] && bounds[static_cast<size_t>(iv) + 1] == codes[i] + 1)) ok = false;
         for (int k = 0; ok && k < nI; ++k)
            if (k != iv && cls_of[static_cast<size_t>(k)] == c) ok = false;
         for (int k = 0; ok && k < ncls; ++k)
            if (k != c && TA(q, k) != 0) ok = false;
         if (ok) {
            q = TA(q, c);
            if (q == 0) ok = false;
         }
      }
      brute_equiv = ok;
   }
   // ---- 7. fast path: <= 8 states per automaton, fused byte tables (one v_perm_b32 per input byte) ---------------------
   // `.match.` runs one forward pass of A over the whole row (api_internal_m.F90:258-302): no R, no candidate list, so the
   // tile kernel applies whenever the tables fit; its prefix/suffix gate is evaluated on the row bytes by the kernel.
   const bool is_match = op == OP_MATCH;
   const bool fast = is_match ? A.n <= 8 : (brute_equiv && A.n <= 8 && R.n <= 8);
   // Symbol ids of the fast tables: 0..127 = the ASCII byte itself; 128+c = a multi-byte (or invalid) character of class c
   // (fx_translate rewrites such bytes); 254 = KILL (all-dead row, feeds the end of a row); 255 = SKIP (continuation byte
   // inside a valid character).
   std::vector<uint8_t> fa(256 * 8, 0), fr(256 * 8, 0);
   if (fast) {
      for (int b = 0; b < 128; ++b)
         for (int s = 0; s < 8; ++s) {
            fa[static_cast<size_t>(b) * 8 + s] = s < A.n ? static_cast<uint8_t>(TA(s, ac[static_cast<size_t>(b)])) : 0;
            fr[static_cast<size_t>(b) * 8 + s] = (!is_match && s < R.n) ? static_cast<uint8_t>(TR(s, ac[static_cast<size_t>(b)])) : static_cast<uint8_t>(0);
         }
      h.flags |= FXP_F_FAST_OK;
      const bool utf8 = (is_match || (r_has_skip && !prefilter)) && ncls <= 126;   // id 254 stays an all-dead row (end-of-row kill symbol)
      if (utf8) {
         h.flags |= FXP_F_FAST_UTF8;
         for (int c = 0; c < ncls; ++c)
            for (int s = 0; s < 8; ++s) {
               fa[static_cast<size_t>(128 + c) * 8 + s] = s < A.n ? static_cast<uint8_t>(TA(s, c)) : 0;
               fr[static_cast<size_t>(128 + c) * 8 + s] = (!is_match && s < R.n) ? static_cast<uint8_t>(TR(s, c)) : static_cast<uint8_t>(0);
            }
      }
      // symbol 255 (SKIP: continuation byte inside a character, also the pad behind rows shorter than their 16-byte chunks):
      // identity for A; for R "same W, hit cleared" (a plain identity when R was built SKIP-blind).  254 (KILL) stays all-dead.
      for (int s = 0; s < 8; ++s) {
         fa[255u * 8 + s] = static_cast<uint8_t>(s < A.n ? s : 0);
         const int sk = (is_match || s >= R.n) ? 0 : (r_has_skip ? R.T[static_cast<size_t>(s) * R.ncol + nI] : s);
         fr[255u * 8 + s] = static_cast<uint8_t>(sk);
      }
      h.flags |= FXP_F_RAGGED_OK;
      int accmin = A.n, hitmin = is_match ? 0 : R.n;
      for (int s = A.n - 1; s >= 0 && A.out[static_cast<size_t>(s)]; --s) accmin = s;
      if (!is_match)
         for (int s = R.n - 1; s >= 0 && R.out[static_cast<size_t>(s)]; --s) hitmin = s;
      h.fast_accA_min = static_cast<uint32_t>(accmin);
      h.fast_hitR_min = static_cast<uint32_t>(hitmin);
      h.fast_R_start = h.R_start;
      h.fast_A_init = is_match ? h.M_start : h.A_init;
      uint64_t fm = 0;
      for (int s = 0; s < A.n && s < 8; ++s)
         if (fin[static_cast<size_t>(s)]) fm |= uint64_t(1) << (8 * s);
      h.fast_finalM[0] = static_cast<uint32_t>(fm);
      h.fast_finalM[1] = static_cast<uint32_t>(fm >> 32);
   }
   h.off_fastA = bl.put(fa.data(), fa.size());
   h.off_fastR = bl.put(fr.data(), fr.size());

   // ---- 8. chain tables: any automaton whose class-indexed tables fit 16-bit row offsets (LDS chain kernel) ---------------
   {
      const uint32_t ncols = static_cast<uint32_t>(ncls) + 3, row_bytes = ncols * 2;
      const uint32_t col_skip = static_cast<uint32_t>(ncls), col_kill = static_cast<uint32_t>(ncls) + 1, col_final = static_cast<uint32_t>(ncls) + 2;
      const bool chain = (is_match || brute_equiv) && !fast && ncls <= 126 && static_cast<uint64_t>(A.n) * row_bytes < 65536u &&
                         (is_match || static_cast<uint64_t>(R.n) * row_bytes < 65536u);
      std::vector<uint16_t> cm(256), ctr, cta;
      if (chain) {
         for (uint32_t sym = 0; sym < 256; ++sym) {
            uint32_t col = col_kill;
            if (sym < 128) col = ac[sym];
            else if (sym == 255) col = col_skip;
            else if (sym - 128 < static_cast<uint32_t>(ncls)) col = sym - 128;
            cm[sym] = static_cast<uint16_t>(2 * col);
         }
         ctr.assign(is_match ? 0 : static_cast<size_t>(R.n) * ncols, 0);
         cta.assign(static_cast<size_t>(A.n) * ncols, 0);
         for (int st = 0; !is_match && st < R.n; ++st) {
            for (int c = 0; c < ncls; ++c) ctr[static_cast<size_t>(st) * ncols + c] = static_cast<uint16_t>(TR(st, c) * row_bytes);
            const int sk = r_has_skip ? R.T[static_cast<size_t>(st) * R.ncol + nI] : st;
            ctr[static_cast<size_t>(st) * ncols + col_skip] = static_cast<uint16_t>(sk * row_bytes);
            ctr[static_cast<size_t>(st) * ncols + col_kill] = static_cast<uint16_t>(st * row_bytes);
         }
         for (int st = 0; st < A.n; ++st) {
            for (int c = 0; c < ncls; ++c) cta[static_cast<size_t>(st) * ncols + c] = static_cast<uint16_t>(TA(st, c) * row_bytes);
            cta[static_cast<size_t>(st) * ncols + col_skip] = static_cast<uint16_t>(st * row_bytes);
            cta[static_cast<size_t>(st) * ncols + col_kill] = 0;
            cta[static_cast<size_t>(st) * ncols + col_final] = fin[static_cast<size_t>(st)];
         }
         int accmin = A.n, hitmin = is_match ? 0 : R.n;
         for (int st = A.n - 1; st >= 0 && A.out[static_cast<size_t>(st)]; --st) accmin = st;
         if (!is_match)
            for (int st = R.n - 1; st >= 0 && R.out[static_cast<size_t>(st)]; --st) hitmin = st;
         h.flags |= FXP_F_CHAIN_OK | FXP_F_RAGGED_OK;
         if (is_match || (r_has_skip && !prefilter)) h.flags |= FXP_F_CHAIN_UTF8;
         h.chain_row_bytes = row_bytes;
         h.chain_R_start = h.R_start * row_bytes;
         h.chain_A_init = (is_match ? h.M_start : h.A_init) * row_bytes;
         h.chain_hit_min = static_cast<uint32_t>(hitmin) * row_bytes;
         h.chain_acc_min = static_cast<uint32_t>(accmin) * row_bytes;
         h.chain_TR_bytes = static_cast<uint32_t>(ctr.size() * 2);
         h.chain_TA_bytes = static_cast<uint32_t>(cta.size() * 2);
      }
      h.off_chain_cls = bl.put(cm.data(), cm.size() * 2);
      h.off_chain_TR = bl.put(ctr.data(), ctr.size() * 2);
      h.off_chain_TA = bl.put(cta.data(), cta.size() * 2);
   }
   return finish(h, bl);
}

Program make_search_literal(const std::string& all) {
   Blob bl;
   bl.b.assign(sizeof(FxpHeader), 0);
   FxpHeader h;
   std::memset(&h, 0, sizeof(h));
   h.mode = FXP_MODE_SEARCH_LITERAL;
   h.len_all = static_cast<uint32_t>(all.size());
   h.off_all = bl.put(all.data(), all.size());
   // Tile-kernel tables for INDEX(str, all): the right-to-left pass runs the KMP automaton of the REVERSED literal over raw
   // bytes; it is in its last state exactly at the first byte of an occurrence, and the last such hit seen is the leftmost
   // occurrence.  No forward pass: to = from + len - 1.  (A literal holding a NUL byte would also match the kernel's NUL
   // sentinels: those stay on the general kernel.)
   const int m = static_cast<int>(all.size());
   std::vector<uint8_t> fa(256 * 8, 0), fr(256 * 8, 0);
   std::vector<uint16_t> cm(256, 0), ctr, cta;
   if (m >= 1 && all.find('\0') == std::string::npos) {
      const std::string rev(all.rbegin(), all.rend());
      std::vector<int> delta(static_cast<size_t>(m + 1) * 256, 0);   // KMP automaton of rev: delta[q][b], state m = full match
      {
         auto P = [&](int j) { return static_cast<int>(static_cast<unsigned char>(rev[static_cast<size_t>(j)])); };
         delta[static_cast<size_t>(P(0))] = 1;
         int x = 0;   // state reached on the longest proper border of rev[0..j)
         for (int j = 1; j < m; ++j) {
            for (int b2 = 0; b2 < 256; ++b2) delta[static_cast<size_t>(j) * 256 + b2] = delta[static_cast<size_t>(x) * 256 + b2];
            delta[static_cast<size_t>(j) * 256 + P(j)] = j + 1;
            x = delta[static_cast<size_t>(x) * 256 + P(j)];
         }
         for (int b2 = 0; b2 < 256; ++b2) delta[static_cast<size_t>(m) * 256 + b2] = delta[static_cast<size_t>(x) * 256 + b2];   // overlapping occurrences
      }
      // byte classes: each distinct literal byte, plus "any other byte"
      std::vector<int> cls(256, -1);
      int ncls = 0;
      for (unsigned char ch : all)
         if (cls[ch] < 0) cls[ch] = ncls++;
      const int other = ncls++;
      int other_byte = -1;
      for (int b = 0; b < 256; ++b)
         if (cls[static_cast<size_t>(b)] < 0) {
            cls[static_cast<size_t>(b)] = other;
            if (other_byte < 0) other_byte = b;
         }
      h.flags |= FXP_F_RAW_BYTES;
      if (all.find('\xFF') == std::string::npos) h.flags |= FXP_F_RAGGED_OK;   // pad byte 0xFF cannot advance the literal's automaton
      h.nR = static_cast<uint32_t>(m + 1);
      if (m + 1 <= 8) {
         for (int b = 0; b < 256; ++b)
            for (int q = 0; q <= m; ++q) fr[static_cast<size_t>(b) * 8 + q] = static_cast<uint8_t>(delta[static_cast<size_t>(q) * 256 + b]);
         h.flags |= FXP_F_FAST_OK;
         h.fast_R_start = 0;
         h.fast_hitR_min = static_cast<uint32_t>(m);
         h.fast_accA_min = 8;
      } else if (ncls <= 126 && static_cast<uint64_t>(m + 1) * (ncls + 3) * 2 < 65536u && other_byte >= 0) {
         const uint32_t ncols = static_cast<uint32_t>(ncls) + 3, row_bytes = ncols * 2;
         for (int b = 0; b < 256; ++b) cm[static_cast<size_t>(b)] = static_cast<uint16_t>(2 * cls[static_cast<size_t>(b)]);
         ctr.assign(static_cast<size_t>(m + 1) * ncols, 0);
         std::vector<int> rep(static_cast<size_t>(ncls), other_byte);
         for (unsigned char ch : all) rep[static_cast<size_t>(cls[ch])] = ch;
         for (int q = 0; q <= m; ++q)
            for (int c = 0; c < ncls; ++c)
               ctr[static_cast<size_t>(q) * ncols + c] = static_cast<uint16_t>(delta[static_cast<size_t>(q) * 256 + rep[static_cast<size_t>(c)]] * row_bytes);
         cta.assign(ncols, 0);   // a single dead row: the forward pass is not used
         h.flags |= FXP_F_CHAIN_OK;
         h.n_classes = static_cast<uint32_t>(ncls);
         h.chain_row_bytes = row_bytes;
         h.chain_R_start = 0;
         h.chain_A_init = 0;
         h.chain_hit_min = static_cast<uint32_t>(m) * row_bytes;
         h.chain_acc_min = 0xFFFFFFFFu;
         h.chain_TR_bytes = static_cast<uint32_t>(ctr.size() * 2);
         h.chain_TA_bytes = static_cast<uint32_t>(cta.size() * 2);
      }
   }
   h.off_fastA = bl.put(fa.data(), fa.size());
   h.off_fastR = bl.put(fr.data(), fr.size());
   h.off_chain_cls = bl.put(cm.data(), cm.size() * 2);
   h.off_chain_TR = bl.put(ctr.data(), ctr.size() * 2);
   h.off_chain_TA = bl.put(cta.data(), cta.size() * 2);
   const uint32_t none = bl.put(nullptr, 0);
   h.off_bounds = h.off_bound_cls = h.off_ascii_cls = h.off_TA = h.off_TR = h.off_accA = h.off_hitR = h.off_finalM = none;
   h.off_prefix = h.off_suffix = h.off_cls_page = h.off_cls_pages = none;
   return finish(h, bl);
}

Program compile(const std::string& pattern, int op, const Limits& lim) {
   std::string buff;
   if (op == OP_SEARCH) {
      buff = f_trim(pattern);   // forgex.F90:95,260
   } else {
      // forgex.F90:182-190 with utility_m.f90:23-53
      std::string adj = f_adjustl(pattern);
      bool caret = !adj.empty() && adj[0] == '^';
      buff = caret ? pattern.substr(1) : pattern;
      std::string tr = f_trim(pattern);
      bool dollar = !tr.empty() && tr[tr.size() - 1] == '$';
      if (dollar) {
         int n = f_len_trim(pattern) - 1;
         if (n < 0) n = 0;
         buff = buff.substr(0, std::min(static_cast<size_t>(n), buff.size()));
      }
   }
   Tree tree;
   tree.build(buff);
   if (!tree.is_valid) return invalid_program(tree.code);
   Literals lit = extract_literal(tree);
   if (op == OP_SEARCH && !f_eq(lit.all, "")) return make_search_literal(lit.all);   // forgex.F90:111-130, :281-307
   Nfa nfa = build_nfa(tree, lim.max_nfa_states);
   return compile_from_nfa(nfa, lit, op, lim);
}

}   // namespace fxc
