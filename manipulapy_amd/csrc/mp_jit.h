// Run-time kernel specialisation (see mp_jit.cpp).
#pragma once

// threads per block of the whole-line inverse-dynamics kernel mp_spec_id_co (its source and its launcher must agree)
#define MP_JIT_ID_CO_BLOCK 64
#include <string>
#include <vector>

#include "mp_model.h"

// The translation unit that gets compiled for `M` (model literal + kernel wrappers).
// part 0: every specialised kernel; part 1: the one-row-per-lane float32 inverse dynamics alone (built with another scheduling strategy)
std::string mp_jit_source(const MpModel<float>& Mf, const MpModel<double>& Md, int part = 0);
// Code object for gfx950, from the disk cache when present.  0 = ok, otherwise `err` holds the hiprtc log.
int mp_jit_compile(const MpModel<float>& Mf, const MpModel<double>& Md, std::vector<char>* code, bool* from_cache,
                   std::string* err, int part = 0);
