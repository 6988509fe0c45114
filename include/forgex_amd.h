/* forgex_amd.h -- C ABI of the MI355X batch regex-match path (libforgex_amd.so).
 *
 * The reference (ShinobuAmasaki/forgex) has no FFI layer: its boundary is the Fortran module `forgex`
 * (reference src/forgex.F90:14-54).  These entry points are what a Fortran `iso_c_binding` interface block
 * for that module binds for the batch match path (see INTEGRATION.md and forgex_amd/fortran/forgex.F90):
 *
 *   fxamd_compile            replaces the per-element  buff=trim(pattern); tree%build; extract_literal;
 *                            automaton%preprocess; automaton%init   (reference src/forgex.F90:95-140, :182-223,
 *                            :260-317) by ONE compile per batch; status codes are the reference's
 *                            src/essential/error_m.F90:12-38 enum values.
 *   fxamd_compile_nfa        same, for a host that keeps its own parser/NFA builder and hands over the
 *                            range-NFA of `nfa_graph_t` (reference src/nfa/nfa_graph_m.F90:24-37) plus the
 *                            literals of `extract_literal` (reference src/ast/syntax_tree_optimize_m.F90:42-55).
 *   fxamd_match_batch_*      replaces the elemental evaluation of `pattern .in. str(:)`, `pattern .match. str(:)`
 *                            and `regex` over a rank-1 character array, i.e. do_matching_including /
 *                            do_matching_exactly (reference src/api_internal_m.F90:31-167, :171-303) for every row.
 *   fxamd_strerror           get_error_message (reference src/essential/error_m.F90:127-211).
 *
 * Rows are the storage of a Fortran `character(row_len) :: s(n)`: n*row_len contiguous bytes, no terminators.
 * All functions return 0 on success or a negative FXAMD_E_* code; they never abort the process.
 * A program handle may be used from several host threads: enqueueing is serialised per handle, the per-call device scratch is
 * kept per (device, stream), so calls that use different streams may overlap on the device; distinct handles are independent.
 * There is NO CPU matching path: every match call needs a HIP device and fails with FXAMD_E_HIP otherwise.
 */
#ifndef FORGEX_AMD_H
#define FORGEX_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fxamd_program fxamd_program;

/* op */
#define FXAMD_OP_SEARCH 0 /* `.in.`, `regex`, `regex_f` */
#define FXAMD_OP_MATCH 1  /* `.match.` */

/* return codes */
#define FXAMD_OK 0
#define FXAMD_E_ARG -1     /* bad argument */
#define FXAMD_E_HIP -2     /* HIP runtime error / no device (fxamd_last_hip_error() has the hipError_t) */
#define FXAMD_E_NOMEM -3
#define FXAMD_E_BLOB -4    /* malformed program blob */
#define FXAMD_E_UNSUPPORTED -5 /* valid pattern the device path cannot run (status >= 100, e.g. DFA state explosion) */

/* Values written to from/to for an INVALID pattern by the regex-style entry (reference forgex.F90:266-274). */
#define FXAMD_INVALID_CHAR_INDEX (-9999)

/* ---- compile (host only; works without a GPU) -------------------------------------------------------- */
/* Identical (op, pattern) pairs share one compiled program: the library keeps the 64 most recently compiled ones, so a loop of
 * scalar calls -- the elemental operators recompile per element -- pays for the compile, the table upload and the device scratch
 * once.  Every handle returned must still be released with fxamd_program_free (it drops a reference). */
int fxamd_compile(const char* pattern, int64_t pattern_len, int op, fxamd_program** out, int32_t* status);

/* Range-NFA hand-over.  States are 1..n_states; transition t goes src[t] -> dst[t] and carries the segments
 * seg_min/seg_max[seg_begin[t] .. seg_begin[t+1]) ; a segment (-1,-1) marks an epsilon transition (SEG_EPSILON,
 * reference src/essential/segment_m.F90:48).  Literals may be empty (len 0). */
int fxamd_compile_nfa(int32_t n_states, int32_t entry, int32_t exit_state, int64_t n_transitions, const int32_t* src,
                      const int32_t* dst, const int64_t* seg_begin, const int32_t* seg_min, const int32_t* seg_max,
                      const char* lit_all, int64_t len_all, const char* lit_prefix, int64_t len_prefix,
                      const char* lit_suffix, int64_t len_suffix, int op, fxamd_program** out, int32_t* status);

void fxamd_program_free(fxamd_program* p);
int32_t fxamd_program_status(const fxamd_program* p);          /* 0 = valid pattern */
int64_t fxamd_program_blob_size(const fxamd_program* p);       /* flattened table image (program.h wire format) */
int fxamd_program_blob(const fxamd_program* p, void* buf, int64_t capacity);
int fxamd_program_from_blob(const void* blob, int64_t size, fxamd_program** out);
/* facts for tests/diagnostics: info[0..7] = mode, flags, nA, nR, n_classes, status, total_bytes, n_bounds */
int fxamd_program_info(const fxamd_program* p, int32_t* info8);
const char* fxamd_strerror(int32_t status);
/* the same message copied (without a terminator) into the caller's buffer; returns its length (<= capacity).  For hosts that
 * cannot turn a C string into their own without an impure intrinsic (Fortran `pure` procedures: c_f_pointer is impure). */
int64_t fxamd_strerror_copy(int32_t status, char* buf, int64_t capacity);

/* ---- match (HIP device required) --------------------------------------------------------------------- */
/* Upload the tables to the current HIP device (idempotent; done lazily by the match calls otherwise). */
int fxamd_program_upload(fxamd_program* p);

/* Upload the tables to the current device and allocate the scratch of (current device, hip_stream) for batches of up to max_rows
 * rows now (4 bytes per row), so that later fxamd_match_batch_device calls on that stream only enqueue kernels (stream capture,
 * latency-sensitive callers).  Optional: the match calls do both lazily. */
int fxamd_program_reserve(fxamd_program* p, int64_t max_rows, void* hip_stream);

/* Device-resident batch: d_rows, d_flags (n bytes: 0/1), d_from, d_to (n int32 each, may both be NULL) are
 * DEVICE pointers; the work is enqueued on `hip_stream` (a hipStream_t, NULL = default stream) and is
 * asynchronous.  `.in.`: flags = verdict, from/to = 1-based byte span of regex() (0,0 when none).
 * `.match.`: flags = verdict, from/to untouched.  Invalid pattern: all flags 0, from/to 0.
 * Any n, row_len and alignment give the same (reference-exact) results; the tile kernels take rows of 2 bytes .. 64 KiB at any base
 * address (16-byte aligned batches load fastest), other shapes run on the general kernel (one lane per row, roughly 20x slower).  The handle keeps its
 * uploaded tables per device and its scratch per (device, stream): the call works on whatever device is current, and calls on
 * one handle may overlap on the device when they use different streams.  n is not limited by the 32-bit row numbers of the kernels'
 * work lists: a batch of more than 2^30 rows is enqueued in slices of that many rows on the same stream. */
int fxamd_match_batch_device(fxamd_program* p, const uint8_t* d_rows, int64_t n, int64_t row_len, uint8_t* d_flags,
                             int32_t* d_from, int32_t* d_to, void* hip_stream);

/* PACKED results -- what a multi-GPU host gathers (one image per shard): 1 bit per row (row i = bit i & 63 of the
 * little-endian 64-bit word i >> 6), then, with spans, from[n] and to[n] narrowed to uint8 (row_len <= 255), uint16 (<= 65535) or
 * int32; the sections start at 16-byte multiples (fxamd_packed_layout: offsets of from / to, total size, bytes per span).
 * d_packed: total_bytes of device memory, 16-byte aligned.  Rows of up to 256 bytes are packed by the search kernel itself (the
 * flag word is the ballot of the wave that owns the tile: 2.125 bytes per row instead of 9 at row_len <= 255); other shapes run
 * the usual pipeline into the handle's scratch and are packed by one more kernel.  `.match.` programs: bits only. */
int fxamd_packed_layout(int64_t n, int64_t row_len, int with_spans, int64_t* off_from, int64_t* off_to, int64_t* total_bytes,
                        int32_t* span_bytes);
int fxamd_match_batch_device_packed(fxamd_program* p, const uint8_t* d_rows, int64_t n, int64_t row_len, int with_spans,
                                    uint8_t* d_packed, void* hip_stream);
/* packed image -> flags u8[n] (0/1), from / to int32[n] (both NULL: flags only); device pointers, asynchronous on the stream */
int fxamd_unpack_results(const uint8_t* d_packed, int64_t n, int64_t row_len, int with_spans, uint8_t* d_flags, int32_t* d_from,
                         int32_t* d_to, void* hip_stream);

/* m patterns against the same device-resident rows (the reference's elemental operators accept an ARRAY of patterns,
 * src/forgex.F90:74, :163): progs[i] fills d_flags[i*n .. i*n+n) (and d_from / d_to likewise).  Patterns whose automata fit the
 * 8-state tile tables share ONE pass over rows of up to 128 bytes (up to 8 patterns per launch, their tables side by side in LDS:
 * the rows are read from HBM once; on longer rows one pipeline per pattern is faster and is what runs); every other pattern runs
 * its own pipeline, enqueued on the same stream.  A handle may appear more
 * than once (identical patterns share one cached program): it is computed once and its results are copied to its other slots. */
int fxamd_match_multi_device(fxamd_program* const* progs, int32_t m, const uint8_t* d_rows, int64_t n, int64_t row_len,
                             uint8_t* d_flags, int32_t* d_from, int32_t* d_to, void* hip_stream);

/* Host-buffer entry used by the Fortran module; synchronous.  The batch flows through two chunk slots (about 64 MB of rows each,
 * own stream, device buffers and pinned result staging, taken from a process-wide pool of such pipes and handed back): the H2D
 * copy of one chunk overlaps the kernels and the D2H copy of the other.  `.match.` programs leave h_from / h_to untouched.
 * Callable from several threads at once, on one handle or many (each call works with its own pipe). */
int fxamd_match_batch_host(fxamd_program* p, const uint8_t* h_rows, int64_t n, int64_t row_len, uint8_t* h_flags,
                           int32_t* h_from, int32_t* h_to);

/* Pin / unpin a caller's host array in place (hipHostRegister): fxamd_match_batch_host then reads the rows by DMA straight from the
 * caller's memory instead of staging pageable memory through the runtime's bounce buffers.  For callers that keep a large batch in
 * one array and match it repeatedly (or once, when the array is large: pinning costs about as much as one pass over pageable memory).
 * The array stays usable as before; unregister before freeing it. */
int fxamd_host_register(void* p, int64_t bytes);
int fxamd_host_unregister(void* p);

/* Free the device scratch (work lists, staging of packed calls, NFA bitsets) that idle programs of the compile cache keep between
 * calls; returns the bytes freed.  The library does so by itself, least recently used first, once the cached programs together keep
 * more than 1 GB (a single program: 512 MB); a host that shares the GPU with another allocator calls this to have the memory back now. */
int64_t fxamd_cache_trim(void);

/* ---- device-resident batches for hosts without a device runtime of their own ---------------------------------------------------
 * (the Fortran module's type(fx_batch); extends the surface of reference src/forgex.F90:24-54 instead of replacing it: the same
 * operators and `regex` accept a batch in place of a character array).  The rows are uploaded once -- or a caller's device pointer is
 * wrapped, not owned -- and stay in HBM across calls and patterns.  fxamd_batch_run enqueues m >= 1 patterns over the rows on the
 * batch's own stream (asynchronous) and leaves the results in device buffers the batch owns: result set i = flags[i*n .. i*n+n) and,
 * with spans, from / to likewise (`.match.` programs: flags only).  fxamd_batch_fetch copies one result set to host arrays,
 * fxamd_batch_count reduces its flags on the device (8 bytes cross the bus), fxamd_batch_results hands out the device pointers.
 * fetch / count / sync are synchronous; the entries of one batch are serialised.
 * Stream ordering: a batch works on a PRIVATE non-blocking stream (fxamd_batch_results hands it out) on the device its rows live on.
 * Nothing orders that stream against the caller's: wrapped rows must be complete when fxamd_batch_run is called -- or the caller says
 * which stream produces them: fxamd_batch_after(b, stream) makes everything enqueued on `stream` so far happen before the batch's next
 * work (an event recorded on `stream`, waited for on the batch's stream; also how a consumer that still reads the result buffers of the
 * previous run is put before the next one).  Results handed out by fxamd_batch_results are complete after fxamd_batch_sync, or for work
 * the caller orders behind the batch's stream itself. */
typedef struct fxamd_batch fxamd_batch;
int fxamd_batch_upload(const uint8_t* h_rows, int64_t n, int64_t row_len, fxamd_batch** out);
int fxamd_batch_wrap(const uint8_t* d_rows, int64_t n, int64_t row_len, fxamd_batch** out);
void fxamd_batch_free(fxamd_batch* b);
int fxamd_batch_info(const fxamd_batch* b, int64_t* n, int64_t* row_len);
int fxamd_batch_run(fxamd_program* const* progs, int32_t m, fxamd_batch* b, int with_spans);
int fxamd_batch_sync(fxamd_batch* b);
int fxamd_batch_after(fxamd_batch* b, void* producer_hip_stream);
int fxamd_batch_fetch(fxamd_batch* b, int32_t which, uint8_t* h_flags, int32_t* h_from, int32_t* h_to);
int fxamd_batch_count(fxamd_batch* b, int32_t which, int64_t* n_matches);
int fxamd_batch_results(fxamd_batch* b, const uint8_t** d_flags, const int32_t** d_from, const int32_t** d_to, int32_t* sets, void** hip_stream);
void fxamd_f_batch_upload(const uint8_t* h_rows, int64_t n, int64_t row_len, fxamd_batch** out, int32_t* rc);
void fxamd_f_batch_wrap(const uint8_t* d_rows, int64_t n, int64_t row_len, fxamd_batch** out, int32_t* rc);
void fxamd_f_batch_free(fxamd_batch* b, int32_t* rc);
void fxamd_f_batch_run(fxamd_program* const* progs, int32_t m, fxamd_batch* b, int32_t with_spans, int32_t* rc);
void fxamd_f_batch_sync(fxamd_batch* b, int32_t* rc);
void fxamd_f_batch_fetch(fxamd_batch* b, int32_t which, uint8_t* h_flags, int32_t* h_from, int32_t* h_to, int32_t* rc);
void fxamd_f_batch_count(fxamd_batch* b, int32_t which, int64_t* n_matches, int32_t* rc);

/* Subroutine forms of four entries above for Fortran `pure` hosts (forgex_amd/fortran/forgex.F90 binds these): a PURE FUNCTION may
 * only have INTENT(IN) / VALUE dummies (F2018 C1590; gfortran rejects the function forms in a pure interface) and a compiler may
 * merge or drop pure-function calls, so every output -- the return code included -- is a pointer argument here.  Same semantics. */
void fxamd_f_compile(const char* pattern, int64_t pattern_len, int op, fxamd_program** out, int32_t* status, int32_t* rc);
void fxamd_f_program_free(fxamd_program* p, int32_t* rc);
void fxamd_f_strerror_copy(int32_t status, char* buf, int64_t capacity, int64_t* n);
void fxamd_f_match_batch_host(fxamd_program* p, const uint8_t* h_rows, int64_t n, int64_t row_len, uint8_t* h_flags,
                              int32_t* h_from, int32_t* h_to, int32_t* rc);

/* Which kernel path the last fxamd_match_batch_device call on this handle used (tests / diagnostics).  Multi-pass pipeline of
 * fx_search_fast / fx_match_fast (rows longer than 256 bytes, `.match.`, FXAMD_MULTIPASS=1): 1 = first pass + decode pass over deferred
 * tiles, 3 = first pass + general fix-up over a worklist, 5 / 6 = the same for automata with more than 8 states, 7 = byte-level tables
 * over every tile + fix-up of structurally invalid rows, 8 = first pass, byte-level tables on deferred tiles, same fix-up.
 * 2 = general kernel, 4 = NFA state-set simulation (DFA too large to build).
 * One-launch kernel fx_search_one: 9 = class-level tables + in-LDS decode, 10 = per-tile selection of class-level / byte-level tables,
 * 11 = byte-level tables on every tile (12 / 13 / 14: the same with the general row procedure for queued rows).
 * 15 = first pass shared with other patterns (fx_search_multi).  16 = 256-byte rows: half-row first pass + ONE gated follow-up of the
 * one-launch kernel over the tiles that pass left.  17 = `.match.` and the `.in.` verdict (no spans) over rows of 2 to 32 bytes:
 * fx_match_tiny / fx_search_tiny (a lane takes a span of 64 / L whole rows) + the gated row-level fix-up of rows with bytes >= 0x80
 * (a stream under hipGraph capture keeps path 9-14).
 * 18 = searches WITH SPANS over rows of 2 to 128 bytes on the 8-state tables: fx_search_span (a lane owns a 128-byte span of 128 / RL whole
 * rows, RL = the row length rounded up to 16 / 32 / 64 / 128; plain or packed results) + ONE gated follow-up of the one-launch kernel over
 * the tiles it marked (a byte >= 0x80, a row in the overlap state of a bordered prefix); for candidate-list driver programs that follow-up
 * runs the general row procedure on queued rows (still 18).  20 = the same kernel on the NIBBLE tables as the first pass of the multi-pass
 * pipeline (9..16-state programs, rows of exactly 16 / 32 / 64 / 128 bytes, plain results), followed by that pipeline's gated passes.
 * Both keep the one-launch kernel on a stream under hipGraph capture.  256-byte rows of programs with more than 8 states (chain and nibble tables) report 5 / 6 / 8 as well: the same pipeline with a half-row first pass at four waves per SIMD (one-launch
 * kernel on a stream under hipGraph capture); rows longer than 256 bytes on the chain tables: 7 or 5 / 6, walked in 128-byte segments.
 * (Environment hooks for the tests: FXAMD_NO_BYTE_DFA, FXAMD_NO_W16, FXAMD_MULTIPASS, FXAMD_NO_HALF, FXAMD_NO_MULTI, FXAMD_NO_CACHE,
 * FXAMD_FORCE_GENERAL, FXAMD_NO_A8, FXAMD_NO_SPEC, FXAMD_NO_TINY, FXAMD_NO_ADAPT, FXAMD_MULTI_NO_BYTES, FXAMD_MULTI_INQ, FXAMD_MULTI_SERIAL,
 * FXAMD_MULTI_W16 (fxamd_match_multi_device: 9..16-state programs join the shared first pass on their nibble tables; off by default: measured no faster),
 * FXAMD_NO_SPAN (paths 18 / 20 off: the one-launch kernel), FXAMD_SPAN_LENS (bit mask, default 111: bits 0..3 = rows that get 128 / 64 / 32 / 16
 * bytes of LDS take path 18; bit 4 = candidate-list driver programs too at rows of up to 64 bytes; bit 5 = ragged rows, any length 2..127 that
 * is not one of the four; bit 6 = path 20),
 * FXAMD_NO_PACK_FIRST (packed results through fx_pack instead of straight from the first passes),
 * FXAMD_NO_LATCH (the half-row first pass and the span kernel walk the plain format of the reverse automaton, not its latched one: program.h FXP_F_R_LATCH),
 * FXAMD_HALF_SCH (bit s: table scheme s -- 0 v_perm, 1 chain, 2 nibble -- takes half rows; bit 3: 128-byte segments of long chain rows;
 * bit 4: 128-byte rows on the chain tables in 64-byte halves);
 * grid experiments: FXAMD_ONE_GRID, FXAMD_ONE_ROUND_MB, FXAMD_ONE_BLOCKS, FXAMD_HALF_ROUNDS.  They are read once
 * per process; `fxamd_reload_env` of forgex_amd_bench.h reads them again.) */
int fxamd_last_path(const fxamd_program* p);
int fxamd_last_hip_error(void);
int fxamd_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
