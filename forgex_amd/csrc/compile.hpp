// Host-side table compiler: pattern (or an externally built range-NFA) -> flattened match program.
// Replaces, for the batch path, the per-element work of reference src/forgex.F90:95-140
// (trim / tree%build / extract_literal / automaton%preprocess / automaton%init) by ONE compile per batch.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "frontend.hpp"
#include "program.h"

namespace fxc {

enum Op : int {
   OP_SEARCH = 0,   // `.in.`, `regex`, `regex_f`   (pattern is TRIMmed: forgex.F90:95,260)
   OP_MATCH = 1,    // `.match.`                     (leading ^ / trailing $ stripped textually: forgex.F90:182-190)
};

struct Limits {
   int max_nfa_states = 8192;
   int max_dfa_states = 4096;   // per automaton; the reference's own hard limit is 16385 lazily-built states
};

struct Program {
   int status = 0;                 // fxfe::Status; 0 = valid pattern
   std::vector<uint8_t> blob;      // FxpHeader + tables (see program.h)
   const FxpHeader& hdr() const { return *reinterpret_cast<const FxpHeader*>(blob.data()); }
};

// pattern exactly as the Fortran caller passed it (untrimmed)
Program compile(const std::string& pattern, int op, const Limits& lim = Limits());

// the text the parser sees: trim(pattern) for `.in.` / regex (forgex.F90:95,260); leading ^ / trailing $ stripped for `.match.` (:182-190)
std::string pattern_text(const std::string& pattern, int op);

// Lower-level entry for an integrating host that keeps its own parser/NFA builder:
// the NFA of reference `nfa_graph_t` after `build_nfa_graph`, plus the three literals of `extract_literal`.
Program compile_from_nfa(const fxfe::Nfa& nfa, const fxfe::Literals& lit, int op, const Limits& lim = Limits());

// wire format: FNV-1a checksum of an image (its checksum field read as zero) and the structural check fxamd_program_from_blob
// applies to images from outside (0 = sound; every table inside the image, every index entry inside its table)
uint32_t blob_checksum(const uint8_t* blob, size_t size);
int validate_blob(const uint8_t* blob, size_t size);

// whole-pattern literal for `.in.` / regex: raw-byte INDEX (forgex.F90:111-130, :281-307)
Program make_search_literal(const std::string& all);

}   // namespace fxc
