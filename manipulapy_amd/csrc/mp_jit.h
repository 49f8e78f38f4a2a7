// Run-time kernel specialisation (see mp_jit.cpp).
#pragma once
#include <string>
#include <vector>

#include "mp_model.h"

// The translation unit that gets compiled for `M` (model literal + kernel wrappers).
std::string mp_jit_source(const MpModel<float>& Mf, const MpModel<double>& Md);
// Code object for gfx950, from the disk cache when present.  0 = ok, otherwise `err` holds the hiprtc log.
int mp_jit_compile(const MpModel<float>& Mf, const MpModel<double>& Md, std::vector<char>* code, bool* from_cache,
                   std::string* err);
