// TEST-ONLY harness: compiles the product's table compiler and its per-row match procedure
// (forgex_amd/csrc/row_engine.hpp, the body of the general HIP kernel) for the HOST, so that table and
// driver logic can be checked against the oracle in the CPU-only container.  It is not part of the
// product library and is never used as a fallback: libforgex_amd.so has no host matching path.
// Speaks the I/M/R/V protocol of oracle/ref_driver.f90 (answers "U <status>" for patterns the GPU build
// does not support, e.g. DFA state explosion).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "../../forgex_amd/csrc/compile.hpp"
#include "../../forgex_amd/csrc/row_engine.hpp"

namespace {
struct HostRow {
   const uint8_t* p;
   uint32_t operator[](int j) const { return p[j]; }
};
long g_byte_rows = 0, g_byte_exceptions = 0;   // FX_HW_BYTES=1: rows answered by the byte tables / sent on to the decode path
void run_any(const fxrow::ProgView& pv, const fxc::Program& p, const HostRow& r, int L, fxrow::Result& res) {
   static const bool use_bytes = std::getenv("FX_HW_BYTES") != nullptr;
   if (use_bytes && (p.hdr().flags & FXP_F_BYTE_DFA)) {   // what the tile kernels' BYTES modes compute; exceptions fall through
      static const bool w16 = std::getenv("FX_HW_BYTES") && std::string(std::getenv("FX_HW_BYTES")) == "w16";   // the 16-state v_perm format
      int rc = fxrow::byte_tables_row(p.blob.data(), r, L, res, w16);
      if (rc == -1 && w16) rc = fxrow::byte_tables_row(p.blob.data(), r, L, res, false);
      if (rc == 0) {
         ++g_byte_rows;
         return;
      }
      if (rc == 1) ++g_byte_exceptions;
   }
   // FX_HW_FAST=1: programs that carry tile-kernel tables are searched the way those kernels search pure-ASCII rows -- by brute
   // force, prefilter literals ignored (the compile-time equivalence proof under test)
   static const bool as_fast = std::getenv("FX_HW_FAST") != nullptr;
   bool force_brute = false;
   if (as_fast && (p.hdr().flags & (FXP_F_FAST_OK | FXP_F_CHAIN_OK | FXP_F_W16_OK)) && (p.hdr().flags & FXP_F_PREFILTER) && L >= 1) {
      force_brute = true;
      for (int j = 0; j < L; ++j) force_brute = force_brute && r[j] < 0x80u;
      if (force_brute) ++g_byte_rows;
   }
   if (p.hdr().flags & FXP_F_NFA_SIM) {   // bitset simulation of NFA state sets (DFA too large)
      std::vector<uint32_t> scratch(2 * p.hdr().nfa_words);
      fxrow::NfaSim sim(pv, scratch.data());
      fxrow::run_row(pv, sim, r, L, res);
   } else {
      fxrow::DfaSim sim(pv);
      fxrow::run_row(pv, sim, r, L, res, force_brute);
   }
}
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

#pragma GCC visibility push(default)   // built with -fvisibility=hidden: the checker exports its C entry points only
extern "C" {
// returns status; writes flag/from/to for ONE row
int hw_run(const char* pat, int64_t plen, int op, const uint8_t* row, int64_t L, int32_t* flag, int32_t* from, int32_t* to) {
   fxc::Program p = fxc::compile(std::string(pat, static_cast<size_t>(plen)), op);
   if (p.status != 0) return p.status;
   fxrow::ProgView pv(p.blob.data());
   HostRow r{row};
   fxrow::Result res;
   run_any(pv, p, r, static_cast<int>(L), res);
   *flag = static_cast<int32_t>(res.flag);
   *from = res.from;
   *to = res.to;
   return 0;
}
// batch with one compile (what the GPU path does)
int hw_batch(const char* pat, int64_t plen, int op, const uint8_t* rows, int64_t n, int64_t L, uint8_t* flags, int32_t* from,
             int32_t* to) {
   fxc::Program p = fxc::compile(std::string(pat, static_cast<size_t>(plen)), op);
   if (p.status != 0) return p.status;
   fxrow::ProgView pv(p.blob.data());
   for (int64_t i = 0; i < n; ++i) {
      HostRow r{rows + i * L};
      fxrow::Result res;
      run_any(pv, p, r, static_cast<int>(L), res);
      flags[i] = static_cast<uint8_t>(res.flag);
      if (from) from[i] = res.from;
      if (to) to[i] = res.to;
   }
   return 0;
}
// the same on `nthreads` host threads (full-batch parity checks of the GPU results: bench.py, tests/test_gpu_parity.py)
int hw_batch_mt(const char* pat, int64_t plen, int op, const uint8_t* rows, int64_t n, int64_t L, uint8_t* flags, int32_t* from,
                int32_t* to, int nthreads) {
   fxc::Program p = fxc::compile(std::string(pat, static_cast<size_t>(plen)), op);
   if (p.status != 0) return p.status;
   if (nthreads < 1) nthreads = 1;
   auto work = [&](int64_t lo, int64_t hi) {
      fxrow::ProgView pv(p.blob.data());
      for (int64_t i = lo; i < hi; ++i) {
         HostRow r{rows + i * L};
         fxrow::Result res;
         run_any(pv, p, r, static_cast<int>(L), res);
         flags[i] = static_cast<uint8_t>(res.flag);
         if (from) from[i] = res.from;
         if (to) to[i] = res.to;
      }
   };
   std::vector<std::thread> th;
   for (int t = 0; t < nthreads; ++t) th.emplace_back(work, n * t / nthreads, n * (t + 1) / nthreads);
   for (auto& t : th) t.join();
   return 0;
}
// wire-format check of the program this compiler emits for a pattern (0 = sound), and of an arbitrary image
int hw_validate(const char* pat, int64_t plen, int op) {
   fxc::Program p = fxc::compile(std::string(pat, static_cast<size_t>(plen)), op);
   return fxc::validate_blob(p.blob.data(), p.blob.size());
}
int hw_validate_blob(const uint8_t* blob, int64_t size) { return fxc::validate_blob(blob, static_cast<size_t>(size)); }
uint32_t hw_blob_checksum(const uint8_t* blob, int64_t size) { return fxc::blob_checksum(blob, static_cast<size_t>(size)); }

// The range-NFA and the literals the front end builds for a pattern, flattened the way INTEGRATION.md section B describes for
// the reference's nfa_graph_t -- the input of fxamd_compile_nfa.  Two calls: with cap == 0 the counts are returned in
// counts[0..5] = n_states, entry, exit, n_transitions, n_segments, status; with cap >= counts the arrays are filled.
int hw_dump_nfa(const char* pat, int64_t plen, int op, int64_t* counts, int64_t cap_tr, int64_t cap_seg, int32_t* src, int32_t* dst,
                int64_t* seg_begin, int32_t* seg_min, int32_t* seg_max, char* lits, int64_t lit_cap, int64_t* lit_len) {
   const std::string buff = fxc::pattern_text(std::string(pat, static_cast<size_t>(plen)), op);
   fxfe::Tree tree;
   tree.build(buff);
   counts[5] = tree.is_valid ? 0 : tree.code;
   if (!tree.is_valid) return 1;
   fxfe::Literals lit = fxfe::extract_literal(tree);
   fxfe::Nfa nfa = fxfe::build_nfa(tree, 8192);
   counts[5] = nfa.status;
   if (nfa.status != 0) return 1;
   int64_t nt = 0, ns = 0;
   for (int i = 1; i <= nfa.nfa_top; ++i)
      for (const fxfe::NfaTransition& tr : nfa.nodes[static_cast<size_t>(i)].forward) {
         if (tr.dst == fxfe::NFA_NULL_TRANSITION) continue;
         if (nt < cap_tr) {
            src[nt] = i;
            dst[nt] = tr.dst;
            seg_begin[nt] = ns;
         }
         for (const fxfe::Seg& sg : tr.c) {
            if (ns < cap_seg) {
               seg_min[ns] = sg.min;
               seg_max[ns] = sg.max;
            }
            ++ns;
         }
         ++nt;
      }
   if (nt <= cap_tr && cap_tr > 0) seg_begin[nt] = ns;
   counts[0] = nfa.nfa_top;
   counts[1] = nfa.entry;
   counts[2] = nfa.exit;
   counts[3] = nt;
   counts[4] = ns;
   const std::string* ls[3] = {&lit.all, &lit.prefix, &lit.suffix};
   int64_t used = 0;
   for (int k = 0; k < 3; ++k) {
      lit_len[k] = static_cast<int64_t>(ls[k]->size());
      if (used + lit_len[k] <= lit_cap) std::memcpy(lits + used, ls[k]->data(), ls[k]->size());
      used += lit_len[k];
   }
   return 0;
}

// symbol-id image of ONE row (length L, multiple of 16) as fx_translate produces it; `expect` gets the same image derived
// from the strict forward parse (fwd_symbol).  Returns 0, or status / -1 when the program has no UTF-8 fast tables.
int hw_translate(const char* pat, int64_t plen, const uint8_t* row, int64_t L, uint8_t* got, uint8_t* expect) {
   fxc::Program p = fxc::compile(std::string(pat, static_cast<size_t>(plen)), fxc::OP_SEARCH);
   if (p.status != 0) return p.status;
   const FxpHeader& h = p.hdr();
   if (!(h.flags & FXP_F_FAST_UTF8) || (L & 15)) return -1;
   const uint8_t* b = p.blob.data();
   fxrow::ClassTables ct{reinterpret_cast<const uint16_t*>(b + h.off_cls_page), reinterpret_cast<const uint16_t*>(b + h.off_cls_pages),
                         reinterpret_cast<const uint16_t*>(b + h.off_bound_cls), reinterpret_cast<const int32_t*>(b + h.off_bounds), h.n_bounds};
   const uint32_t sym_ffff = 128u + h.cls_ffff;
   const int CH = static_cast<int>(L / 16);
   for (int k = 0; k < CH; ++k) {
      uint32_t w[6] = {0, 0, 0, 0, 0, 0};
      if (k > 0) std::memcpy(&w[0], row + 16 * k - 4, 4);
      std::memcpy(&w[1], row + 16 * k, 16);
      if (k + 1 < CH) std::memcpy(&w[5], row + 16 * k + 16, 4);
      fxrow::Cell16 o = fxrow::translate_cell16(w[0], w[1], w[2], w[3], w[4], w[5], ct, sym_ffff);
      std::memcpy(got + 16 * k, &o, 16);
   }
   fxrow::ProgView pv(b);
   HostRow r{row};
   int j = 0;
   while (j < L) {
      int next;
      uint32_t cls = fxrow::fwd_symbol(pv, r, static_cast<int>(L), j, next);
      expect[j] = row[j] < 0x80 ? row[j] : static_cast<uint8_t>(128u + cls);
      for (int q = j + 1; q < next; ++q) expect[q] = 255;
      j = next;
   }
   return 0;
}

// byte-level tables of a pattern: info[0..4] = present, byte classes, states of A, states of R, table bytes
void hw_byte_info(const char* pat, int64_t plen, int op, int32_t* info) {
   fxc::Program p = fxc::compile(std::string(pat, static_cast<size_t>(plen)), op);
   const FxpHeader& h = p.hdr();
   info[0] = (h.flags & FXP_F_BYTE_DFA) ? 1 : 0;
   info[1] = static_cast<int32_t>(h.byte_n_classes);
   info[2] = h.byte_row_bytes ? static_cast<int32_t>(h.byte_TA_bytes / h.byte_row_bytes) : 0;
   info[3] = h.byte_row_bytes ? static_cast<int32_t>(h.byte_TR_bytes / h.byte_row_bytes) : 0;
   info[4] = static_cast<int32_t>(h.byte_TA_bytes + h.byte_TR_bytes);
}
void hw_byte_stats(long* rows, long* exceptions) {
   *rows = g_byte_rows;
   *exceptions = g_byte_exceptions;
}
// program facts for tests: fills info[0..7] = mode, flags, nA, nR, n_classes, status, total_bytes, n_bounds
void hw_info(const char* pat, int64_t plen, int op, int32_t* info) {
   fxc::Program p = fxc::compile(std::string(pat, static_cast<size_t>(plen)), op);
   const FxpHeader& h = p.hdr();
   info[0] = static_cast<int32_t>(h.mode);
   info[1] = static_cast<int32_t>(h.flags);
   info[2] = static_cast<int32_t>(h.nA);
   info[3] = static_cast<int32_t>(h.nR);
   info[4] = static_cast<int32_t>(h.n_classes);
   info[5] = p.status;
   info[6] = static_cast<int32_t>(h.total_bytes);
   info[7] = static_cast<int32_t>(h.n_bounds);
}
}

#ifdef HW_MAIN
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
      int op = f[0] == "M" ? fxc::OP_MATCH : fxc::OP_SEARCH;
      int32_t flag = 0, from = 0, to = 0;
      int st = 0;
      if (f[0] == "V") {
         fxc::Program pr = fxc::compile(pat, fxc::OP_SEARCH);
         std::printf("V %c\n", (pr.status == 0 || pr.status >= 100) ? 'T' : 'F');
         std::fflush(stdout);
         continue;
      }
      st = hw_run(pat.data(), static_cast<int64_t>(pat.size()), op, reinterpret_cast<const uint8_t*>(txt.data()),
                  static_cast<int64_t>(txt.size()), &flag, &from, &to);
      if (st >= 100) {
         std::printf("U %d\n", st);
      } else if (f[0] == "I" || f[0] == "M") {
         std::printf("%s %c\n", f[0].c_str(), (st == 0 && flag) ? 'T' : 'F');
      } else if (f[0] == "R") {
         if (st != 0) {
            std::printf("R -9999 -9999 0 %d -\n", st);
         } else {
            bool m = from > 0 && to > 0;
            std::string sub = m ? txt.substr(static_cast<size_t>(from - 1), static_cast<size_t>(to - from + 1)) : std::string();
            std::printf("R %d %d %d 0 %s\n", m ? from : 0, m ? to : 0, m ? to - from + 1 : 0, tohex(sub).c_str());
         }
      } else {
         std::puts("E bad-op");
      }
      std::fflush(stdout);
   }
   if (std::getenv("FX_HW_STATS")) std::fprintf(stderr, "host_walk: %ld rows answered by the byte tables or by forced brute force, %ld exceptions\n", g_byte_rows, g_byte_exceptions);
   return 0;
}
#endif
