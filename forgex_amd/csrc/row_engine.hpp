// Per-row match procedure over a compiled program (program.h).  This is the body of the GENERAL HIP
// kernel (one wavefront lane = one row): it is written once, as templates over a row accessor, and is
// compiled for the device by kernels.hip.  tests/support/host_walk.cpp compiles the same text for the
// host as a TEST-ONLY harness (so table/driver logic can be checked without a GPU); the product library
// never contains a host instantiation -- there is no CPU fallback.
//
// Reference behaviour reproduced here (file:line of the reference):
//   search driver        src/api_internal_m.F90:31-167   (NUL wrapping, restart order, span arithmetic)
//   full-match driver    src/api_internal_m.F90:171-303  (prefix/suffix gate, leading-NUL retry, verdict)
//   candidate list       src/essential/utility_m.f90:58-117
//   literal fast paths   src/forgex.F90:111-130, :207-213
//   UTF-8 stepping       src/essential/utf8_m.f90:44-140, :168-246, :338-430 (structural validity only;
//                        every byte of an invalid sequence is one U+FFFF symbol: api_internal_m.F90:129-133)
#pragma once
#include <stdint.h>

#include "program.h"

#if defined(__HIPCC__)
#define FX_HD __device__ __forceinline__
#else
#define FX_HD inline
#endif

namespace fxrow {

// View of a program blob with every table pointer and scalar resolved ONCE (the blob may live in LDS or in global
// memory; all reads below are single loads, no header indirection in the per-symbol loops).
struct ProgView {
   const uint8_t* base;
   const FxpHeader* hd;
   const int32_t* bounds;
   const uint16_t *bound_cls_p, *ascii_p, *ta_p, *tr_p, *page_p, *pages_p;
   const uint8_t *finalM_p, *prefix_p, *suffix_p, *all_p;
   uint32_t ncls, n_bounds, cls_nul_v, cls_ffff_v;
   FX_HD explicit ProgView(const uint8_t* b) : base(b) {
      hd = reinterpret_cast<const FxpHeader*>(b);
      const FxpHeader hh = *hd;
      bounds = reinterpret_cast<const int32_t*>(b + hh.off_bounds);
      bound_cls_p = reinterpret_cast<const uint16_t*>(b + hh.off_bound_cls);
      ascii_p = reinterpret_cast<const uint16_t*>(b + hh.off_ascii_cls);
      ta_p = reinterpret_cast<const uint16_t*>(b + hh.off_TA);
      tr_p = reinterpret_cast<const uint16_t*>(b + hh.off_TR);
      page_p = reinterpret_cast<const uint16_t*>(b + hh.off_cls_page);
      pages_p = reinterpret_cast<const uint16_t*>(b + hh.off_cls_pages);
      finalM_p = b + hh.off_finalM;
      prefix_p = b + hh.off_prefix;
      suffix_p = b + hh.off_suffix;
      all_p = b + hh.off_all;
      ncls = hh.n_classes;
      n_bounds = hh.n_bounds;
      cls_nul_v = hh.cls_nul;
      cls_ffff_v = hh.cls_ffff;
   }
   FX_HD const FxpHeader& h() const { return *hd; }
   FX_HD uint32_t ascii_cls(uint32_t b) const { return ascii_p[b]; }
   FX_HD uint32_t TA(uint32_t s, uint32_t c) const { return ta_p[s * ncls + c]; }
   FX_HD uint32_t TR(uint32_t s, uint32_t c) const { return tr_p[s * ncls + c]; }
   FX_HD uint32_t finalM(uint32_t s) const { return finalM_p[s]; }
   FX_HD uint32_t prefix(uint32_t i) const { return prefix_p[i]; }
   FX_HD uint32_t suffix(uint32_t i) const { return suffix_p[i]; }
   FX_HD uint32_t all(uint32_t i) const { return all_p[i]; }
   FX_HD uint32_t class_of_code(int32_t code) const {
      if (code < 0x10000) {   // BMP: page table (two reads)
         const uint32_t pg = page_p[static_cast<uint32_t>(code) >> 6];
         return pages_p[pg * 64u + (static_cast<uint32_t>(code) & 63u)];
      }
      uint32_t lo = 0, hi = n_bounds;   // last interval whose first code point is <= code
      while (hi - lo > 1) {
         uint32_t mid = (lo + hi) >> 1;
         if (bounds[mid] <= code) lo = mid;
         else hi = mid;
      }
      return bound_cls_p[lo];
   }
};

// ---- automaton back ends: the row procedures below only see "init / step / alive" ---------------------------------------
struct DfaSim {   // dense tables T_A / T_R (the normal case)
   const ProgView& pv;
   uint32_t st;
   FX_HD explicit DfaSim(const ProgView& p) : pv(p), st(0) {}
   FX_HD void fwd_init() { st = pv.h().A_init; }
   FX_HD bool fwd_step(uint32_t cls) {   // returns "destination accepts"
      const uint32_t e = pv.TA(st, cls);
      st = e & FXP_STATE_MASK;
      return (e & FXP_FLAG_BIT) != 0;
   }
   FX_HD bool alive() const { return st != 0; }
   FX_HD void rev_init() { st = pv.h().R_start; }
   FX_HD bool rev_step(uint32_t cls) {   // returns "a non-empty match starts at this symbol"
      const uint32_t e = pv.TR(st, cls);
      st = e & FXP_STATE_MASK;
      return (e & FXP_FLAG_BIT) != 0;
   }
   FX_HD void match_init() { st = pv.h().M_start; }
   FX_HD bool match_final() { return st != 0 && pv.finalM(st) != 0; }
   FX_HD bool at_overlap_sink() const { return st == pv.h().R_inv; }   // after a reverse pass (FXP_F_OVERLAP_SINK)
};

struct NfaSim {   // FXP_F_NFA_SIM: state SETS as bitsets in per-row scratch memory (DFA too large to build)
   const ProgView& pv;
   uint32_t *a, *b;   // current / next set, nfa_words words each
   uint32_t words, N1, entry, exit_s;
   const uint32_t *fwd, *rev, *init, *f0, *rstart;
   FX_HD NfaSim(const ProgView& p, uint32_t* scratch) : pv(p) {
      const FxpHeader hh = *p.hd;
      words = hh.nfa_words;
      N1 = hh.nfa_N + 1;
      entry = hh.nfa_entry;
      exit_s = hh.nfa_exit;
      a = scratch;
      b = scratch + words;
      fwd = reinterpret_cast<const uint32_t*>(p.base + hh.off_nfa_fwd);
      rev = reinterpret_cast<const uint32_t*>(p.base + hh.off_nfa_rev);
      init = reinterpret_cast<const uint32_t*>(p.base + hh.off_nfa_init);
      f0 = reinterpret_cast<const uint32_t*>(p.base + hh.off_nfa_f0);
      rstart = reinterpret_cast<const uint32_t*>(p.base + hh.off_nfa_rstart);
   }
   FX_HD bool at_overlap_sink() const { return false; }   // (no such state in the NFA simulation)
   FX_HD void load(const uint32_t* src) {
      for (uint32_t i = 0; i < words; ++i) a[i] = src[i];
   }
   FX_HD bool test(uint32_t s) const { return (a[s >> 5] >> (s & 31u)) & 1u; }
   FX_HD void step(const uint32_t* table, uint32_t cls) {   // b = union over members x of a of table[cls][x]; swap
      for (uint32_t i = 0; i < words; ++i) b[i] = 0;
      for (uint32_t wi = 0; wi < words; ++wi) {
         uint32_t bits = a[wi];
         while (bits) {
            const uint32_t low = bits & (0u - bits);
            uint32_t k = 0;
            for (uint32_t t = low; t > 1u; t >>= 1) ++k;
            bits ^= low;
            const uint32_t* row = table + (static_cast<size_t>(cls) * N1 + (wi * 32u + k)) * words;
            for (uint32_t i = 0; i < words; ++i) b[i] |= row[i];
         }
      }
      uint32_t* t = a;
      a = b;
      b = t;
   }
   FX_HD void fwd_init() { load(init); }
   FX_HD bool fwd_step(uint32_t cls) {
      step(fwd, cls);
      return test(exit_s);
   }
   FX_HD bool alive() const {
      uint32_t any = 0;
      for (uint32_t i = 0; i < words; ++i) any |= a[i];
      return any != 0;
   }
   FX_HD void rev_init() { load(rstart); }
   FX_HD bool rev_step(uint32_t cls) {
      step(rev, cls);
      const bool hit = test(entry);
      for (uint32_t i = 0; i < words; ++i) a[i] |= f0[i];
      return hit;
   }
   FX_HD void match_init() {   // api_internal_m.F90:280-289: consume the leading NUL if the initial state can, else skip it
      load(init);
      step(fwd, pv.cls_nul_v);
      if (!alive()) load(init);
   }
   FX_HD bool match_final() {   // accept at ci = n+2, or after the trailing NUL at n+3
      if (!alive()) return false;
      if (test(exit_s)) return true;
      step(fwd, pv.cls_nul_v);
      return test(exit_s);
   }
};

FX_HD bool is_cont(uint32_t b) { return (b >> 6) == 2u; }
FX_HD int lead_len(uint32_t b) {
   if ((b >> 5) == 6u) return 2;
   if ((b >> 4) == 14u) return 3;
   if ((b >> 3) == 30u) return 4;
   return 0;
}
template <class Row>
FX_HD int32_t decode(const Row& r, int i, int n) {   // utf8_m.f90:338-430 (arithmetic, no range checks)
   uint32_t b0 = r[i];
   uint32_t code = n == 2 ? (b0 & 0x1Fu) : (n == 3 ? (b0 & 0x0Fu) : (b0 & 0x07u));
   for (int k = 1; k < n; ++k) code = (code << 6) | (r[i + k] & 0x3Fu);
   return static_cast<int32_t>(code);
}

// symbol that STARTS at 0-based text index j (a character start); L = text length
template <class Row>
FX_HD uint32_t fwd_symbol(const ProgView& pv, const Row& r, int L, int j, int& next) {
   uint32_t b = r[j];
   if (b < 0x80u) {
      next = j + 1;
      return pv.ascii_cls(b);
   }
   int n = lead_len(b);
   if (n != 0 && j + n <= L) {
      bool ok = true;
      for (int k = 1; k < n; ++k) ok = ok && is_cont(r[j + k]);
      if (ok) {
         next = j + n;
         return pv.class_of_code(decode(r, j, n));
      }
   }
   next = j + 1;
   return pv.cls_ffff_v;
}

// symbol that ENDS at 0-based text index j, given that j+1 is a character start (or the end of the text)
template <class Row>
FX_HD uint32_t back_symbol(const ProgView& pv, const Row& r, int j, int& start) {
   uint32_t b = r[j];
   start = j;
   if (b < 0x80u) return pv.ascii_cls(b);
   if (is_cont(b)) {
      int n = 0;
      if (j >= 1 && lead_len(r[j - 1]) == 2) n = 2;
      else if (j >= 2 && is_cont(r[j - 1]) && lead_len(r[j - 2]) == 3) n = 3;
      else if (j >= 3 && is_cont(r[j - 1]) && is_cont(r[j - 2]) && lead_len(r[j - 3]) == 4) n = 4;
      if (n != 0) {
         start = j - n + 1;
         return pv.class_of_code(decode(r, start, n));
      }
   }
   return pv.cls_ffff_v;
}

// byte i (1-based) of NUL // text // NUL
template <class Row>
FX_HD uint32_t wrapped(const Row& r, int L, int i) {
   return (i == 1 || i == L + 2) ? 0u : r[i - 2];
}

// Longest non-empty match of the anchored automaton started at wrapped index `st` (1 <= st <= L+1).
// Returns max_match exactly as api_internal_m.F90:119-137 computes it (0 = none).
template <class Row, class Sim>
FX_HD int anchored_max_match(const ProgView& pv, Sim& sim, const Row& r, int L, int st) {
   sim.fwd_init();
   int mm = 0;
   int j;   // 0-based text index of the next symbol
   if (st == 1) {
      if (sim.fwd_step(pv.cls_nul_v)) mm = 2;
      j = 0;
   } else {
      j = st - 2;
   }
   while (sim.alive() && j < L) {
      int next;
      const uint32_t cls = fwd_symbol(pv, r, L, j, next);
      const bool acc = sim.fwd_step(cls);
      j = next;
      if (acc) mm = j + 2;
   }
   if (sim.alive()) {   // trailing NUL
      if (sim.fwd_step(pv.cls_nul_v)) mm = L + 3;
   }
   return mm;
}

FX_HD void span_from(int start, int max_match, int L, int& from, int& to) {   // api_internal_m.F90:140-148
   from = start - 1;
   if (from == 0) from = 1;
   to = (max_match >= L + 2) ? L : max_match - 2;
}

// first occurrence (1-based, 0 = none) of pat[0..m) in the wrapped string, searching wrapped indices >= lo
template <class Row, class PatFn>
FX_HD int find_wrapped(const Row& r, int L, PatFn pat, int m, int lo) {
   for (int i = lo; i + m - 1 <= L + 2; ++i) {
      bool ok = true;
      for (int k = 0; ok && k < m; ++k) ok = wrapped(r, L, i + k) == pat(k);
      if (ok) return i;
   }
   return 0;
}

// ---- UTF-8 -> fast-path symbol ids (the decode passes of the tile kernels; also compiled by the test-only host harness) ---
FX_HD uint32_t u8_is_cont(uint32_t b) { return (b & 0xC0u) == 0x80u; }
FX_HD uint32_t u8_lead_len(uint32_t b) { return b >= 0xF8u ? 0u : (b >= 0xF0u ? 4u : (b >= 0xE0u ? 3u : (b >= 0xC0u ? 2u : 0u))); }

// w[0] = last dword of the previous cell of the row (0 at the row start), w[1..4] = this cell, w[5] = first dword of the next cell
struct ClassTables {   // plain pointers only
   const uint16_t *page_p, *pages_p, *bound_cls_p;
   const int32_t* bounds;
   uint32_t n_bounds;
   FX_HD uint32_t class_of_code(int32_t code) const {
      if (code < 0x10000) return pages_p[(uint32_t)page_p[(uint32_t)code >> 6] * 64u + ((uint32_t)code & 63u)];
      uint32_t lo = 0, hi = n_bounds;
      while (hi - lo > 1) {
         const uint32_t mid = (lo + hi) >> 1;
         if (bounds[mid] <= code) lo = mid;
         else hi = mid;
      }
      return bound_cls_p[lo];
   }
};

struct Cell16 {
   uint32_t x, y, z, w;
};
// byte k of the 8-byte pair {hi,lo} shifted right by n bytes (n = 1..3): v_alignbyte_b32 on the device
FX_HD uint32_t fx_alignbyte(uint32_t hi, uint32_t lo, uint32_t n) {
#if defined(__HIPCC__)
   return __builtin_amdgcn_alignbyte(hi, lo, n);
#else
   return static_cast<uint32_t>(((static_cast<uint64_t>(hi) << 32) | lo) >> (8 * n));
#endif
}

// SWAR formulation: the byte-type predicates (continuation, 2/3/4-byte lead) are computed for whole dwords as masks with
// bit 7 of each byte set; "valid lead", "covered continuation" follow from byte-shifted ANDs across the 24-byte window.
// Only the class of valid lead bytes needs per-position work (code point assembly + two table reads).
FX_HD Cell16 translate_cell16(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4, uint32_t w5, const ClassTables& pv,
                              uint32_t sym_ffff) {
   const uint32_t w[6] = {w0, w1, w2, w3, w4, w5};
   if (((w[1] | w[2] | w[3] | w[4]) & 0x80808080u) == 0) return Cell16{w[1], w[2], w[3], w[4]};
   const uint32_t M = 0x80808080u;
   uint32_t C[6], L2[6], L3[6], L4[6];
#pragma unroll
   for (int d = 0; d < 6; ++d) {
      const uint32_t x = w[d], s1 = x << 1, s2 = x << 2, s3 = x << 3, s4 = x << 4;
      C[d] = x & ~s1 & M;
      L2[d] = x & s1 & ~s2 & M;
      L3[d] = x & s1 & s2 & ~s3 & M;
      L4[d] = x & s1 & s2 & s3 & ~s4 & M;
   }
   // V*[d]: a structurally valid 2/3/4-byte character starts at this byte (d = 0..4; the window's last dword only feeds C)
   uint32_t V2[5], V3[5], V4[5];
#pragma unroll
   for (int d = 0; d < 5; ++d) {
      const uint32_t c1 = fx_alignbyte(C[d + 1], C[d], 1), c2 = fx_alignbyte(C[d + 1], C[d], 2), c3 = fx_alignbyte(C[d + 1], C[d], 3);
      V2[d] = L2[d] & c1;
      V3[d] = L3[d] & c1 & c2;
      V4[d] = L4[d] & c1 & c2 & c3;
   }
   uint32_t out[4];
   const uint32_t ffff_rep = sym_ffff * 0x01010101u;
#pragma unroll
   for (int d = 1; d < 5; ++d) {
      const uint32_t valid = V2[d] | V3[d] | V4[d];
      const uint32_t va1 = V2[d - 1] | V3[d - 1] | V4[d - 1];
      const uint32_t v34p = V3[d - 1] | V4[d - 1], v34 = V3[d] | V4[d];
      // covered continuation byte: a valid lead 1 byte before, a valid 3/4-byte lead 2 before, or a valid 4-byte lead 3 before
      const uint32_t cov = fx_alignbyte(valid, va1, 3) | fx_alignbyte(v34, v34p, 2) | fx_alignbyte(V4[d], V4[d - 1], 1);
      const uint32_t hi = w[d] & M;
      const uint32_t inval = hi & ~cov & ~valid;
      const uint32_t ff_hi = (hi >> 7) * 0xFFu, ff_cov = (cov >> 7) * 0xFFu, ff_inv = (inval >> 7) * 0xFFu;
      out[d - 1] = (w[d] & ~ff_hi) | ff_cov | (ff_inv & ffff_rep);   // valid lead bytes are patched below
   }
   // no valid lead byte in this cell (ASCII plus broken bytes, e.g. Latin-1 text): nothing to classify -- on the device the whole
   // wave skips the code point assembly when that holds for all of its rows
   if (((V2[1] | V3[1] | V4[1]) | (V2[2] | V3[2] | V4[2]) | (V2[3] | V3[3] | V4[3]) | (V2[4] | V3[4] | V4[4])) == 0)
      return Cell16{out[0], out[1], out[2], out[3]};
   // classes of the valid lead bytes: code point assembly + BMP page lookup, all 16 positions independent
   uint32_t code[16];
#pragma unroll
   for (int q = 0; q < 16; ++q) {
      auto B = [&](int i) -> uint32_t {   // byte at cell offset i, 0 <= i < 20
         const int j = i + 4;
         return (w[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
      };
      const int d = 1 + (q >> 2), sh = (q & 3) * 8 + 7;
      const uint32_t is2 = (V2[d] >> sh) & 1u, is3 = (V3[d] >> sh) & 1u, is4 = (V4[d] >> sh) & 1u;
      const uint32_t b0 = B(q), b1 = B(q + 1) & 0x3Fu, b2 = B(q + 2) & 0x3Fu, b3 = B(q + 3) & 0x3Fu;
      const uint32_t c2 = ((b0 & 0x1Fu) << 6) | b1;
      const uint32_t c3 = ((((b0 & 0x0Fu) << 6) | b1) << 6) | b2;
      const uint32_t c4 = ((((((b0 & 0x07u) << 6) | b1) << 6) | b2) << 6) | b3;
      code[q] = is2 ? c2 : (is3 ? c3 : (is4 ? c4 : 0xFFFFFFFFu));   // 0xFFFFFFFF: not a valid lead, no lookup wanted
   }
   uint32_t pg[16];
#pragma unroll
   for (int q = 0; q < 16; ++q) pg[q] = pv.page_p[code[q] < 0x10000u ? (code[q] >> 6) : 0u];
#pragma unroll
   for (int q = 0; q < 16; ++q) {
      uint32_t c = pv.pages_p[pg[q] * 64u + (code[q] & 63u)];
      if (code[q] != 0xFFFFFFFFu && code[q] >= 0x10000u) c = pv.class_of_code((int32_t)code[q]);   // beyond the BMP (rare)
      const uint32_t patch = code[q] != 0xFFFFFFFFu ? (128u + c) : 0u;
      out[q >> 2] |= patch << ((q & 3) * 8);
   }
   return Cell16{out[0], out[1], out[2], out[3]};
}

// `.match.` gate on the raw row bytes (forgex.F90:207-213, api_internal_m.F90:199-233), shared by the tile kernel and the
// byte-table walk below: 2 = verdict is TRUE, 0 = verdict is FALSE, 1 = the automaton decides
template <class RowFn>
FX_HD uint32_t match_gate(const FxpHeader* h, const uint8_t* prog, const RowFn& row, uint32_t L) {
   const uint32_t lp = h->len_prefix, ls = h->len_suffix, la = h->len_all;
   if ((h->flags & FXP_F_MATCH_LITERAL) && L == la) {
      bool eq = true;
      for (uint32_t k = 0; k < la; ++k) eq = eq && row(k) == prog[h->off_all + k];
      return eq ? 2u : 0u;
   }
   if (lp > 0 && lp == L) {
      bool eq = true;
      for (uint32_t k = 0; k < lp; ++k) eq = eq && row(k) == prog[h->off_prefix + k];
      if (eq) return 2u;
   }
   if (lp > L || ls > L) return 0u;
   bool ok = true;
   if (h->flags & FXP_F_PREFILTER)
      for (uint32_t k = 0; k < lp; ++k) ok = ok && row(k) == prog[h->off_prefix + k];
   if (h->flags & FXP_F_HAS_SUFFIX)
      for (uint32_t k = 0; k < ls; ++k) ok = ok && row(L - ls + k) == prog[h->off_suffix + k];
   return ok ? 1u : 0u;
}

// FXP_F_PREFIX_CHECK (compile.cpp): is the brute-force start s (wrapped index: 1 = the leading NUL, j + 2 = text byte j) one the reference's candidate list tries
// before any other start that could match?  The prefix literal stands at text byte j = s - 2 (wholly inside the text: it holds no NUL) and no occurrence of it
// starts in the lp - 1 bytes before j (utility_m.f90:94-116 collects NON-overlapping occurrences left to right).  `pre(k)`: byte k of the literal.
template <class Row, class Pre>
FX_HD bool prefix_start_ok(const Pre& pre, const int lp, const Row& r, const int L, const int s) {
   if (s < 2) return false;
   const int j = s - 2;
   if (j + lp > L) return false;
   for (int k = 0; k < lp; ++k)
      if (r[j + k] != pre(k)) return false;
   for (int d = 1; d < lp && d <= j; ++d) {   // an occurrence at j - d overlaps the one at j
      bool occ = true;
      for (int k = 0; occ && k < lp; ++k) occ = r[j - d + k] == pre(k);
      if (occ) return false;
   }
   return true;
}

// ... and when the start is NOT such a candidate: does the prefix literal occur in the row at all?  If it occurs nowhere the reference itself searches by brute
// force, suffix literal ignored (api_internal_m.F90:76-81: INDEX of the prefix = 0) -- the tables' answer stands.  This is the common case for the "prefix"
// literals the reference derives from patterns like `(}[abc]){2}\d*c{2,}` (`}}`) or `(\t{3}[a-z]){2}` (six tabs): no match ever contains them.
template <class Row, class Pre>
FX_HD bool prefix_occurs(const Pre& pre, const int lp, const Row& r, const int L) {
   const uint32_t p0 = pre(0);
   for (int i = 0; i + lp <= L; ++i) {
      if (r[i] != p0) continue;
      int k = 1;
      while (k < lp && r[i + k] == pre(k)) ++k;
      if (k == lp) return true;
   }
   return false;
}
// FXP_F_SUFFIX_CHECK: the match from the start s (wrapped index >= 2) with max_match mm (wrapped index of the symbol behind it; >= L + 2: it ran to the row's end)
// ends with the suffix literal, and that occurrence starts at least one byte behind the match's start.
template <class Row, class Suf>
FX_HD bool suffix_end_ok(const Suf& suf, const int ls, const Row& r, const int L, const int s, const int mm) {
   if (s < 2 || mm <= 0) return false;
   const int j = s - 2, e = mm >= L + 2 ? L : mm - 2;   // text bytes [j, e)
   if (e - ls < j + 1) return false;
   for (int k = 0; k < ls; ++k)
      if (r[e - ls + k] != suf(k)) return false;
   return true;
}

struct Result {
   uint32_t flag;   // verdict of `.in.` / `.match.`
   int32_t from, to;   // regex(): 1-based byte span, 0/0 when there is none
};

// force_brute: ignore the prefilter literals and search by brute force -- what the tile kernels do for programs whose candidate-list
// search is PROVEN equal to it (compile.cpp, `brute_equiv`); the test harness uses it to check that proof against the oracle
template <class Row, class Sim>
FX_HD void search_engine(const ProgView& pv, Sim& sim, const Row& r, int L, Result& out, bool force_brute = false) {
   const FxpHeader& h = pv.h();
   out.flag = 0;
   out.from = 0;
   out.to = 0;
   // api_internal_m.F90:68-74 -- `len(string) <= 1 .and. string == ''` : empty text, or ONE blank
   if (L == 0 || (L == 1 && r[0] == 0x20u)) {
      if (h.flags & FXP_F_INIT_ACCEPTING) out.flag = 1;   // ACCEPTED_EMPTY: .in. is true, regex() returns '' / 0 / 0
      return;
   }
   bool brute = force_brute || !(h.flags & FXP_F_PREFILTER);
   int first = 0, suf_idx = -1;
   const int lp = static_cast<int>(h.len_prefix), ls = static_cast<int>(h.len_suffix);
   auto pre = [&](int k) { return pv.prefix(static_cast<uint32_t>(k)); };
   auto suf = [&](int k) { return pv.suffix(static_cast<uint32_t>(k)); };
   if (!brute) {
      // get_index_list_forward, utility_m.f90:58-117 (first element only; the rest is generated on the fly below)
      int idx = find_wrapped(r, L, pre, lp, 1);
      if (ls == 0) {
         suf_idx = L + 3;   // INDEX(text, '', back=.true.) == len(text)+1
      } else {
         suf_idx = -1;
         for (int i = L + 2 - ls + 1; i >= 1; --i) {
            bool ok = true;
            for (int k = 0; ok && k < ls; ++k) ok = wrapped(r, L, i + k) == suf(k);
            if (ok) {
               suf_idx = i;
               break;
            }
         }
      }
      if (idx > 0 && (suf_idx < 0 || idx <= suf_idx)) first = idx;
      if (idx == 0 && (h.flags & FXP_F_PREFIX_NECESSARY)) {
         // the prefix occurs nowhere: the reference falls back to brute force here (api_internal_m.F90:79-81), which on a pure-ASCII
         // row cannot find anything either (every match begins with the prefix) -- skip it.  (Rows with bytes >= 0x80 may spell
         // the prefix in an overlong form that only the automaton sees: they take the fallback.)
         bool ascii = true;
         for (int j = 0; j < L; ++j) ascii = ascii && r[j] < 0x80u;
         if (ascii) return;
      }
      if (first == 0) brute = true;   // api_internal_m.F90:79-81
   }
   if (brute) {
      int s = 0;
      bool restart_loop = !(h.flags & FXP_F_HAS_R);
      if (!restart_loop) {
         sim.rev_init();
         int j = L - 1;
         while (j >= 0) {
            int start;
            const uint32_t cls = back_symbol(pv, r, j, start);
            if (sim.rev_step(cls)) s = start + 2;
            j = start - 1;
         }
         if (sim.rev_step(pv.cls_nul_v)) s = 1;
         if ((h.flags & FXP_F_OVERLAP_SINK) && sim.at_overlap_sink()) {
            if (force_brute) {
               // (test harness: what the tile kernels do with such a row -- they leave it to this procedure WITHOUT force_brute)
               search_engine(pv, sim, r, L, out, false);
               return;
            }
            // Round 6: the reference's OWN fallback to brute force (the prefix occurs nowhere byte-wise, api_internal_m.F90:79-81) on a row whose SYMBOLS hold
            // two overlapping prefix occurrences -- possible only through non-canonical encodings (`αα[^a]*` over α, an overlong α, α): R was composed with the
            // overlap detector and stopped recording hits once it entered the absorbing state, so its last hit is not the leftmost start.  The restart loop
            // with A alone is exact.  (Found by fuzz_prefilter.py FX_FUZZ_UTF8=1 seed 2; the bug dates from round 3's overlap sink.)
            restart_loop = true;
            s = 0;
         }
      }
      if (!restart_loop) {
         if (s == 0) return;
         bool prefix_nowhere = false;   // (the reference's own brute-force fallback: the suffix literal is not consulted either)
         if (force_brute && (h.flags & FXP_F_PREFIX_CHECK) && !prefix_start_ok(pre, lp, r, L, s)) {
            prefix_nowhere = !prefix_occurs(pre, lp, r, L);
            if (!prefix_nowhere) {
               // (test harness: what the tile kernels do with such a row -- they leave it to this procedure WITHOUT force_brute)
               search_engine(pv, sim, r, L, out, false);
               return;
            }
         }
         int mm = anchored_max_match(pv, sim, r, L, s);
         if (force_brute && !prefix_nowhere && (h.flags & FXP_F_SUFFIX_CHECK) && !suffix_end_ok(suf, ls, r, L, s, mm)) {   // (test harness: as above)
            search_engine(pv, sim, r, L, out, false);
            return;
         }
         span_from(s, mm, L, out.from, out.to);
      } else {
         // bounded restart loop, api_internal_m.F90:108-155
         int start = 1;
         while (start < L + 2) {
            int mm = anchored_max_match(pv, sim, r, L, start);
            if (mm > 0) {
               span_from(start, mm, L, out.from, out.to);
               s = start;
               break;
            }
            if (start == 1) start = 2;
            else {
               int next;
               (void)fwd_symbol(pv, r, L, start - 2, next);
               start = next + 2;
            }
         }
         if (s == 0) return;
      }
   } else {
      // candidate-list driver, api_internal_m.F90:84-164
      int suf2 = -1;
      if (h.flags & FXP_F_HAS_SUFFIX) {
         suf2 = 0;
         for (int i = L - ls + 1; i >= 1; --i) {   // INDEX(string, suffix, back=.true.) on the UNWRAPPED text
            bool ok = true;
            for (int k = 0; ok && k < ls; ++k) ok = r[i - 1 + k] == suf(k);
            if (ok) {
               suf2 = i;
               break;
            }
         }
         if (suf2 == 0) return;
      }
      int cand = first;                 // index_list(i)
      int offset = first + lp - 1;      // end of the last collected occurrence
      bool more = true;                 // list construction still running
      int start = (first == 2) ? 1 : first;
      bool at_nul = (first == 2);
      int found = 0, mm = 0;
      while (start < L + 2) {
         if (suf2 >= 0 && suf2 < start) break;
         mm = anchored_max_match(pv, sim, r, L, start);
         if (mm > 0) {
            found = start;
            break;
         }
         if (at_nul) {
            at_nul = false;
            start = cand;
            continue;
         }
         // next element of the index list
         if (!more || !(offset < L + 2)) break;
         int nxt = find_wrapped(r, L, pre, lp, offset + 1);
         if (nxt == 0) break;
         cand = nxt;
         offset = nxt + lp - 1;
         if (suf_idx >= 0 && offset > suf_idx) more = false;
         start = cand;
      }
      if (found == 0) return;
      span_from(found, mm, L, out.from, out.to);
   }
   if (out.from > 0 && out.to > 0) {
      out.flag = 1;
   } else {   // forgex.F90:152-156, :337-343
      out.from = 0;
      out.to = 0;
   }
}

template <class Row>
FX_HD void search_literal(const ProgView& pv, const Row& r, int L, Result& out) {   // forgex.F90:111-130, :281-307
   const int m = static_cast<int>(pv.h().len_all);
   out.flag = 0;
   out.from = 0;
   out.to = 0;
   for (int i = 0; i + m <= L; ++i) {
      bool ok = true;
      for (int k = 0; ok && k < m; ++k) ok = r[i + k] == pv.all(static_cast<uint32_t>(k));
      if (ok) {
         out.flag = 1;
         out.from = i + 1;
         out.to = i + m;
         return;
      }
   }
}

template <class Row, class Sim>
FX_HD void match_engine(const ProgView& pv, Sim& sim, const Row& r, int L, Result& out) {   // forgex.F90:207-226 + api_internal_m.F90:171-303
   const FxpHeader& h = pv.h();
   out.flag = 0;
   out.from = 0;
   out.to = 0;
   const int n = L, lp = static_cast<int>(h.len_prefix), ls = static_cast<int>(h.len_suffix);
   if ((h.flags & FXP_F_MATCH_LITERAL) && n == static_cast<int>(h.len_all)) {
      bool eq = true;
      for (int k = 0; eq && k < n; ++k) eq = r[k] == pv.all(static_cast<uint32_t>(k));
      out.flag = eq ? 1u : 0u;
      return;
   }
   if (n > 0 && lp > 0 && lp == n) {   // :200-205 "prefix == whole string => true"
      bool eq = true;
      for (int k = 0; eq && k < n; ++k) eq = r[k] == pv.prefix(static_cast<uint32_t>(k));
      if (eq) {
         out.flag = 1;
         return;
      }
   }
   if (lp > n || ls > n) return;
   const bool empty_pre = !(h.flags & FXP_F_PREFILTER), empty_post = !(h.flags & FXP_F_HAS_SUFFIX);
   bool matches_pre = true, matches_post = true;
   if (n > 0) {
      if (!empty_pre)
         for (int k = 0; matches_pre && k < lp; ++k) matches_pre = r[k] == pv.prefix(static_cast<uint32_t>(k));
      if (!empty_post)
         for (int k = 0; matches_post && k < ls; ++k) matches_post = r[n - ls + k] == pv.suffix(static_cast<uint32_t>(k));
   } else {
      matches_pre = lp == 0;
      matches_post = ls == 0;
   }
   if (!((empty_pre || matches_pre) && (empty_post || matches_post))) return;
   if (n == 0) {
      out.flag = (h.flags & FXP_F_INIT_ACCEPTING) ? 1u : 0u;
      return;
   }
   sim.match_init();
   int j = 0;
   while (sim.alive() && j < L) {
      int next;
      const uint32_t cls = fwd_symbol(pv, r, L, j, next);
      (void)sim.fwd_step(cls);
      j = next;
   }
   out.flag = sim.match_final() ? 1u : 0u;
}

template <class Row, class Sim>
FX_HD void run_row(const ProgView& pv, Sim& sim, const Row& r, int L, Result& out, bool force_brute = false) {
   switch (pv.h().mode) {
      case FXP_MODE_SEARCH_ENGINE: search_engine(pv, sim, r, L, out, force_brute); break;
      case FXP_MODE_SEARCH_LITERAL: search_literal(pv, r, L, out); break;
      case FXP_MODE_MATCH_ENGINE: match_engine(pv, sim, r, L, out); break;
      default:
         out.flag = 0;
         out.from = 0;
         out.to = 0;
         break;
   }
}

// Byte-level chain tables (FXP_F_BYTE_DFA) walked one byte at a time: the per-row meaning of the tile kernels' BYTES modes
// (backward pass of R over the raw bytes for the leftmost start, forward pass of A for the longest end; `.match.`: one
// forward pass and the FINAL column).  Returns 0 = `out` is the row's result, 1 = the row must be redone by the decode
// path (structurally invalid UTF-8), -1 = the program has no byte tables / the row is outside the tile kernels' domain.
// `w16`: walk the 16-state nibble format of the same automata instead (states are plain ids; 16 nibbles per byte value)
template <class Row>
FX_HD int byte_tables_row(const uint8_t* base, const Row& r, int L, Result& out, bool w16 = false) {
   const FxpHeader* h = reinterpret_cast<const FxpHeader*>(base);
   if (!(h->flags & FXP_F_BYTE_DFA) || L < 1 || (L == 1 && r[0] == 0x20u)) return -1;   // (empty / single-blank text: api_internal_m.F90:68-74)
   if (w16 && !(h->flags & FXP_F_BYTE_W16)) return -1;
   const uint16_t* cmap = reinterpret_cast<const uint16_t*>(base + h->off_byte_cls);
   const uint8_t *TRp = base + (w16 ? h->off_bw16R : h->off_byte_TR), *TAp = base + (w16 ? h->off_bw16A : h->off_byte_TA);
   auto step = [&](const uint8_t* T, uint32_t st, uint32_t byte) -> uint32_t {
      if (w16) return (uint32_t)(T[byte * 8u + (st >> 1)] >> ((st & 1u) * 4u)) & 15u;
      return *reinterpret_cast<const uint16_t*>(T + st + cmap[byte]);
   };
   const uint32_t A_init = w16 ? h->bw16_A_init : h->byte_A_init, R_start = w16 ? h->bw16_R_start : h->byte_R_start;
   const uint32_t hit_min = w16 ? h->bw16_hit_min : h->byte_hit_min, acc_min = w16 ? h->bw16_acc_min : h->byte_acc_min;
   const uint32_t inv_R = w16 ? h->bw16_inv_R : h->byte_inv_R;
   out.flag = 0;
   out.from = 0;
   out.to = 0;
   if (h->mode == FXP_MODE_MATCH_ENGINE) {
      auto row = [&](uint32_t j) -> uint32_t { return r[static_cast<int>(j)]; };
      const uint32_t gate = match_gate(h, base, row, static_cast<uint32_t>(L));
      uint32_t st = A_init;
      for (int j = 0; j < L; ++j) st = step(TAp, st, r[j]);
      const uint32_t fin = w16 ? reinterpret_cast<const uint8_t*>(h->bw16_finalM)[st & 15u]
                               : *reinterpret_cast<const uint16_t*>(TAp + st + 2u * (h->byte_n_classes + 2u));
      if (gate == 2u) out.flag = 1;
      else if (gate == 0u) out.flag = 0;
      else if (st != 0 && fin == 2u) return 1;
      else out.flag = (st != 0 && fin == 1u) ? 1u : 0u;
      return 0;
   }
   uint32_t state = R_start, s = 0;
   for (int j = L - 1; j >= 0; --j) {
      state = step(TRp, state, r[j]);
      if (state >= hit_min) s = static_cast<uint32_t>(j) + 2u;
   }
   state = step(TRp, state, 0u);   // leading NUL
   if (state >= hit_min) s = 1;
   if (state == inv_R) return 1;
   if (s == 0) return 0;
   uint32_t cur = A_init, mm = 0, j = s >= 2 ? s - 2 : 0;
   if (s == 1) {
      cur = step(TAp, cur, 0u);
      mm = cur >= acc_min ? 2u : 0u;
   }
   while (cur != 0 && j <= static_cast<uint32_t>(L)) {   // position L holds the trailing NUL, later positions kill the state
      cur = step(TAp, cur, j < static_cast<uint32_t>(L) ? r[static_cast<int>(j)] : 0u);
      if (cur >= acc_min) mm = j + 3u;
      ++j;
   }
   if (mm != 0) {   // api_internal_m.F90:140-148
      int32_t fr = static_cast<int32_t>(s) - 1;
      if (fr == 0) fr = 1;
      const int32_t tt = mm >= static_cast<uint32_t>(L) + 2u ? L : static_cast<int32_t>(mm) - 2;
      if (fr > 0 && tt > 0) {
         out.flag = 1;
         out.from = fr;
         out.to = tt;
      }
   }
   return 0;
}

}   // namespace fxrow
