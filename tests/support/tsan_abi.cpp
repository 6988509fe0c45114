// TEST INFRASTRUCTURE (CPU): drives the host side of the C ABI from several threads at once -- the compile cache (shared, refcounted programs
// behind one mutex), program images out and in, the cache trim -- in a ThreadSanitizer build of fxamd.hip's host code + front end + table
// compiler (tests/test_host_logic.py::test_c_abi_host_side_under_thread_sanitizer builds both).  No device is needed: nothing here enqueues work.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "forgex_amd.h"

int main() {
   const std::vector<std::string> pats = {"[a-z]+\\d+", "foo(bar|baz)", "\\d{3}-\\d{4}", "[α-ωぁ-ん]+", "^abc$", "a{2,}[xy]", "(ab|cd)+e?", "x(", "[z-a]",
                                          "--[a-z]+", "abc.*xyz", ".*a(a|b){40}c", "\\s+\\S", "q.{10,}z"};
   constexpr int NT = 8, ROUNDS = 40;
   std::atomic<int> errors{0};
   std::vector<std::vector<int32_t>> status(NT, std::vector<int32_t>(pats.size() * 2, -1));
   auto worker = [&](int tid) {
      std::vector<uint8_t> blob;
      for (int r = 0; r < ROUNDS; ++r) {
         for (size_t i = 0; i < pats.size(); ++i) {
            for (int op = 0; op < 2; ++op) {   // FXAMD_OP_SEARCH / FXAMD_OP_MATCH
               const std::string& p = pats[(i + (size_t)tid) % pats.size()];
               fxamd_program* h = nullptr;
               int32_t st = -1;
               const int rc = fxamd_compile(p.data(), (int64_t)p.size(), op, &h, &st);
               if (rc != 0 || h == nullptr) {
                  errors++;
                  continue;
               }
               int32_t& seen = status[tid][((i + (size_t)tid) % pats.size()) * 2 + (size_t)op];
               if (seen != -1 && seen != st) errors++;   // the same (op, pattern) always compiles to the same status
               seen = st;
               int32_t info[8];
               if (fxamd_program_info(h, info) != 0) errors++;
               const int64_t sz = fxamd_program_blob_size(h);
               if (st == 0 && sz > 0 && (r % 4) == tid % 4) {   // image out and in again: programs from blobs are never shared
                  blob.resize((size_t)sz);
                  if (fxamd_program_blob(h, blob.data(), sz) != 0) errors++;
                  fxamd_program* h2 = nullptr;
                  if (fxamd_program_from_blob(blob.data(), sz, &h2) != 0 || h2 == nullptr) errors++;
                  else fxamd_program_free(h2);
               }
               (void)fxamd_strerror(st);
               fxamd_program_free(h);
            }
         }
         if ((r & 7) == tid) (void)fxamd_cache_trim();
      }
   };
   std::vector<std::thread> th;
   for (int t = 0; t < NT; ++t) th.emplace_back(worker, t);
   for (auto& t : th) t.join();
   for (int t = 1; t < NT; ++t)
      for (size_t k = 0; k < status[t].size(); ++k)
         if (status[t][k] != -1 && status[0][k] != -1 && status[t][k] != status[0][k]) errors++;
   std::printf("%s %d\n", errors.load() == 0 ? "THREADS OK" : "THREADS FAILED", errors.load());
   return errors.load() == 0 ? 0 : 1;
}
