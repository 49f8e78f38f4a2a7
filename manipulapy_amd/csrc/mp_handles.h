// Handle layouts shared by the translation units that implement the C ABI (mp_capi.cpp, mp_cpu.cpp).
#pragma once
#include <cstdint>

#include "mp_model.h"

struct mp_model {
  MpModel<double> d;   // n <= MP_MAX_DOF: the unrolled kernels take these by value.  d.n is ALWAYS the joint count.
  MpModel<float> f;
  bool big = false;    // MP_MAX_DOF < n <= MP_BIG_DOF: only bd / bf hold the joints; the looped kernels (csrc/mp_dyn.h) read them
  MpBigModel<double> bd;
  MpBigModel<float> bf;
  uint64_t uid;  // never reused, so a context's device copies cannot alias a destroyed model
};

int mp_set_error(int code, const char* msg);  // thread-local message of mp_last_error (mp_capi.cpp; C++ linkage)
