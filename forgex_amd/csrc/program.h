// Flattened, position-independent "match program" shared by the host compiler (compile.cpp), the HIP
// kernels (kernels.hip) and the test-only host walker.  One blob = header + tables; all offsets are in
// bytes from the start of the blob; the blob is uploaded to HBM as-is and staged into LDS by the kernels.
//
// What the tables encode (see DESIGN.md §3 for the derivation):
//   * symbol classes: code points 0..0x1FFFFF partitioned by every segment boundary of the range-NFA
//     (reference src/nfa/nfa_node_m.F90:410-501 `disjoin_nfa`), then columns merged when identical.
//   * A  = forward ANCHORED DFA of the NFA (what `automaton%construct` would lazily build from the
//          initial closure; reference src/automaton_m.F90:333-381).  State 0 = dead (DFA_INVALID_INDEX).
//   * R  = reverse UNANCHORED DFA: scanning a row right-to-left, R is in a "hit" state at position p iff
//          a NON-EMPTY match starts at p -- the set the reference enumerates with its restart loop
//          (reference src/api_internal_m.F90:108-164), in one linear pass.
#pragma once
#include <stdint.h>

#define FXP_MAGIC 0x31505846u /* "FXP1" */
#define FXP_VERSION 17u

enum FxpMode {
   FXP_MODE_INVALID = 0,         // invalid pattern: every row is "no match" (reference forgex.F90:101-104)
   FXP_MODE_SEARCH_ENGINE = 1,   // .in. / regex through the automaton
   FXP_MODE_SEARCH_LITERAL = 2,  // .in. / regex: whole pattern is a literal -> raw byte INDEX (forgex.F90:111-130)
   FXP_MODE_MATCH_ENGINE = 3,    // .match. (forgex.F90:163-231)
};

enum FxpFlags {
   FXP_F_INIT_ACCEPTING = 1u << 0,  // initial DFA state accepts (empty-text rule, api_internal_m.F90:68-74,247-250)
   FXP_F_PREFILTER = 1u << 1,       // prefix literal is not blank -> candidate-list driver (api_internal_m.F90:76-104)
   FXP_F_HAS_SUFFIX = 1u << 2,      // suffix literal is not blank
   FXP_F_FAST_OK = 1u << 3,         // <=8-state byte tables present and brute-force semantics proven equivalent
   FXP_F_HAS_R = 1u << 4,           // reverse DFA present (else: bounded restart loop)
   FXP_F_MATCH_LITERAL = 1u << 5,   // .match.: `all` literal present -> byte equality when lengths agree (forgex.F90:207-213)
   FXP_F_FAST_UTF8 = 1u << 6,       // fast tables also hold 128+class and SKIP rows: the fast kernel decodes UTF-8 in place
   FXP_F_NFA_SIM = 1u << 7,         // DFA too large: NFA state sets are simulated on the device (bitsets), both directions
   FXP_F_CHAIN_OK = 1u << 8,        // class-indexed LDS chain tables present (automata too large for the v_perm tables)
   FXP_F_CHAIN_UTF8 = 1u << 9,      // ... and they tell SKIP apart: the chain kernel's second pass may decode UTF-8
   FXP_F_BYTE_DFA = 1u << 12,       // byte-level chain tables present (UTF-8 composed into the automata)
   FXP_F_W16_OK = 1u << 13,         // 16-state nibble tables present (automata with 9..16 states: a 64-bit shift per byte instead of the LDS chain)
   FXP_F_W16_UTF8 = 1u << 14,       // ... and they hold the 128+class / SKIP rows (the decode pass may use them)
   FXP_F_BYTE_W16 = 1u << 15,       // the byte-level automata also exist in the 16-state nibble format
   FXP_F_BYTE_A8 = 1u << 18,        // searches: the byte-level FORWARD automaton has <= 8 states and also exists in the v_perm format (b8A)
   FXP_F_SPEC_FWD = 1u << 19,       // ... and a walk of it from the row's first character decides the leftmost start by itself when it finds a match
                                    // (the leading NUL is no start, no state survives U+FFFF, no candidate-list driver): the tile kernels' speculative pass
   FXP_F_PREFIX_CHECK = 1u << 21,   // searches with a prefix literal that is NOT proven a necessary, border-free beginning of every match: the tile tables still run
                                    // (brute-force semantics) and the kernel checks per ROW that the start it found is one the reference's candidate list would
                                    // have tried first -- the prefix literal stands there and no earlier occurrence overlaps it --, else the row goes to the general
                                    // row procedure (round 6; compile.cpp `prefix_check`, row_engine.hpp prefix_start_ok)
   FXP_F_SUFFIX_CHECK = 1u << 22,   // ... and (with FXP_F_PREFIX_CHECK) a suffix literal that is not proven a necessary ending: the kernel also checks that the match it found
                                    // ends with the suffix literal, at least one byte behind the match's start -- what makes the driver's give-up / cut-off rules
                                    // (api_internal_m.F90:99-116) moot for this row (row_engine.hpp suffix_end_ok)
   FXP_F_R_LATCH = 1u << 23,        // searches whose reverse automaton has <= 4 states (and no overlap state): R also exists in a LATCHED 8-state v_perm format
                                    // (off_fastRL; round 6): states 0..3 as in off_fastR, states 4..7 = the same states once a hit state has been entered since
                                    // the walker last cleared the latch (state & 3).  "Did any of this 8-byte group's states hit?" is then the group's LAST state
                                    // >= 4 -- no running maximum over the group's states (4 of a group's 24 vector instructions in the half-row and span kernels)
   FXP_F_NEEDS_NONASCII = 1u << 20, // searches with byte-level tables: no non-empty match is made of ASCII symbols only -- a row without a byte >= 0x80 holds no match
   FXP_F_OVERLAP_SINK = 1u << 17,   // prefix literal with a border: R carries one absorbing state (R_inv) entered when two prefix occurrences overlap
   FXP_F_PREFIX_NECESSARY = 1u << 16,   // every non-empty match begins with the prefix literal (proven on A): a pure-ASCII row without it cannot match
   FXP_F_RAGGED_OK = 1u << 11,      // symbol 255 is inert at the end of a row: rows whose length is not a multiple of 16 may be padded with it
   FXP_F_RAW_BYTES = 1u << 10,      // literal INDEX search: symbols are raw bytes (no UTF-8 decode, no deferral), hit = occurrence start
};

struct FxpHeader {
   uint32_t magic, version, mode, status, flags;
   uint32_t n_classes, n_bounds, nA, nR;
   uint32_t A_init, R_start, M_start;
   uint32_t cls_nul, cls_ffff;
   uint32_t len_prefix, len_suffix, len_all;
   uint32_t fast_accA_min, fast_hitR_min, fast_R_start, fast_A_init;   // fast tables: state >= *_min <=> accepting / hit
                                                                       // (`.match.` programs: fast_A_init = M_start)
   uint32_t off_bounds;      // int32  [n_bounds]   ascending first code point of each interval (bounds[0] == 0)
   uint32_t off_bound_cls;   // uint16 [n_bounds]   class of each interval
   uint32_t off_ascii_cls;   // uint16 [128]
   uint32_t off_TA;          // uint16 [nA * n_classes]   bit15 = destination accepts
   uint32_t off_TR;          // uint16 [nR * n_classes]   bit15 = destination is a hit state
   uint32_t off_accA;        // uint8  [nA]
   uint32_t off_hitR;        // uint8  [nR]
   uint32_t off_finalM;      // uint8  [nA]   .match. verdict of a state reached after the last text byte
   uint32_t off_prefix, off_suffix, off_all;   // raw bytes
   uint32_t off_fastA;       // uint8 [256][8]   next state of each of 8 states per fast-path symbol id:
   uint32_t off_fastR;       // uint8 [256][8]   0..127 ASCII byte, 128+c character of class c, 255 SKIP
   uint32_t total_bytes;
   uint32_t n_pages;         // distinct 64-code-point pages of the BMP class map
   uint32_t off_cls_page;    // uint16 [1024]          page id of code points [64p, 64p+63], p = cp >> 6 (cp < 0x10000)
   uint32_t off_cls_pages;   // uint16 [n_pages][64]   class of each code point of a page
   // ---- NFA simulation (FXP_F_NFA_SIM): states 1..nfa_N are bit positions; sets are nfa_words 32-bit words ----
   uint32_t nfa_N, nfa_words, nfa_entry, nfa_exit;
   uint32_t off_nfa_init;     // uint32 [words]   epsilon closure of the entry state
   uint32_t off_nfa_f0;       // uint32 [words]   states whose closure holds the exit state
   uint32_t off_nfa_rstart;   // uint32 [words]   reverse scan start set (after the trailing NUL)
   uint32_t off_nfa_fwd;      // uint32 [n_classes][N+1][words]   closed successors of a state on a class
   uint32_t off_nfa_rev;      // uint32 [n_classes][N+1][words]   states that reach z' by (closure, one symbol of the class)
   // ---- LDS chain tables (FXP_F_CHAIN_OK): state = byte offset of its row; row = (n_classes + 3) uint16 entries: the
   //      destination row offset per column (classes 0..n_classes-1, then SKIP, then KILL) and a FINAL column that holds
   //      the `.match.` verdict of the state (off_finalM) ----
   uint32_t chain_row_bytes, chain_R_start, chain_A_init, chain_hit_min, chain_acc_min, chain_TR_bytes, chain_TA_bytes;
   uint32_t off_chain_cls;    // uint16 [256]   2 * column of each fast-path symbol id
   uint32_t off_chain_TR;     // uint16 [nR][n_classes + 3]
   uint32_t off_chain_TA;     // uint16 [nA][n_classes + 3]   row 0 = dead
   uint32_t fast_finalM[2];   // v_perm scheme, `.match.`: byte q = 1 when state q gives a true verdict after the last text byte
   // ---- byte-level chain tables (FXP_F_BYTE_DFA): A and R composed with the UTF-8 decoder (reference utf8_m.f90:338-430
   //      `ichar_utf8`: arithmetic decode of every structurally valid sequence), so the tile kernels walk RAW bytes -- no
   //      decode pass.  Same row layout as the class-level chain tables with byte classes as columns.  A structurally invalid
   //      byte sequence (utf8_m.f90:168-246) leads to the absorbing state `byte_inv_*`; such rows (and `.match.` rows ending
   //      inside a character: FINAL column = 2) are redone by the decode pass, which treats every invalid byte as U+FFFF. ----
   uint32_t byte_n_classes, byte_row_bytes, byte_R_start, byte_A_init, byte_hit_min, byte_acc_min, byte_TR_bytes, byte_TA_bytes;
   uint32_t byte_inv_R, byte_inv_A;   // row offsets of the INVALID states (A: `.match.` programs only, else 0 = dead)
   uint32_t off_byte_cls;    // uint16 [256]   2 * column of each byte value
   uint32_t off_byte_TR;     // uint16 [nRb][byte_n_classes + 3]
   uint32_t off_byte_TA;     // uint16 [nAb][byte_n_classes + 3]   row 0 = dead
   // ---- 16-state NIBBLE tables: per symbol 8 bytes = 16 nibbles, nibble j (bits 4j..4j+3 of the little-endian 64-bit entry) = the
   //      next state of state j; states are plain ids 0..15.  One step = (entry >> 4*state) & 15: a 64-bit shift and a mask
   //      (3 VALU with the shift amount) on ONE ds_read_b64 -- half the LDS bytes of a 16-byte-per-symbol v_perm format, which is
   //      what bounds these automata (tools/ubench/step_rate.hip: 1.7-2x the steps per second on UTF-8-like byte spreads). ----
   uint32_t w16_R_start, w16_A_init, w16_hit_min, w16_acc_min;                               // class-level (fast-path symbol ids)
   uint32_t bw16_R_start, bw16_A_init, bw16_hit_min, bw16_acc_min, bw16_inv_R, bw16_inv_A;   // byte-level (raw bytes)
   uint32_t off_w16A, off_w16R;     // uint8 [256][8]   (16 nibbles per symbol)
   uint32_t off_bw16A, off_bw16R;   // uint8 [256][8]
   uint32_t w16_finalM[4], bw16_finalM[4];   // `.match.`: byte j = verdict of state j after the last text byte (byte-level: 2 = redo by the decode path)
   // ---- FXP_F_BYTE_A8: the byte-level forward automaton of a SEARCH (structure errors lead to its dead state: the backward pass has
   //      sent such rows to the exception path before any forward walk) in the 8-state v_perm format, indexed by the raw byte: the
   //      forward pass of a UTF-8 tile costs one v_perm_b32 per byte instead of the nibble format's three instructions ----
   uint32_t off_b8A;      // uint8 [256][8]
   uint32_t b8_A_init, b8_acc_min;
   uint32_t R_inv;        // FXP_F_OVERLAP_SINK: that state of R (a row that ends its backward pass there is left to the general engine)
   uint32_t off_fastRL;   // FXP_F_R_LATCH: uint8 [256][8], the latched format of off_fastR (fast_R_start is its start state too; base state s is a hit state iff s >= fast_hitR_min)
   uint32_t checksum;     // FNV-1a of the whole image with this field read as zero (fxc::blob_checksum); checked by fxamd_program_from_blob
};

#define FXP_STATE_MASK 0x7FFFu
#define FXP_FLAG_BIT 0x8000u
