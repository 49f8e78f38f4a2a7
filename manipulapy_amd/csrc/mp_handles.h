// Handle layouts shared by the translation units that implement the C ABI (mp_capi.cpp, mp_cpu.cpp).
#pragma once
#include <cstdint>

#include "mp_model.h"

struct mp_model {
  MpModel<double> d;
  MpModel<float> f;
  uint64_t uid;  // never reused, so a context's device copies cannot alias a destroyed model
};

int mp_set_error(int code, const char* msg);  // thread-local message of mp_last_error (mp_capi.cpp; C++ linkage)
