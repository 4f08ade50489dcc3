// api_util.h -- error plumbing shared by the extern "C" translation units.
#pragma once
#include <cstdlib>

#include <hip/hip_runtime.h>

#include "../../include/crispy_hip.h"

namespace crispy {

// Sets the calling thread's last-error message and returns `code`.
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
const char* last_error_cstr();
// CRISPY_OK if `device` is a usable gfx950 device, else an error with `who` in the message.
int check_device(int device, const char* who);
bool device_is_gfx950(int dev);
// Called from a catch (...) handler of an extern "C" entry point: classifies the exception in flight
// (std::bad_alloc / std::length_error -> CRISPY_ERR_OOM, anything else -> CRISPY_ERR_HIP), records a message and
// returns the status.  Never throws (the message buffer is a fixed thread-local array).
int fail_exception(const char* where) noexcept;

// Environment variables the library reads -- two kinds, nothing else calls getenv:
//  * test hooks (test_env): read in every build and listed under "Environment" in include/crispy_hip.h.  They select
//    between forms that give bit-identical results (the tests that use them assert exactly that), never a result;
//  * developer knobs (dev_env): A/B switches for tools/ (ramps, request depths, timelines, tile walkers).  Compiled OUT of
//    the release library -- a host's environment must not steer which kernel form the product runs -- and in with
//    `make dev` (-DCRISPY_DEV_KNOBS, ../libcrispy_hip_dev.so; tools pick it through CRISPY_HIP_LIB).
inline const char* test_env(const char* name) { return std::getenv(name); }
#ifdef CRISPY_DEV_KNOBS
inline const char* dev_env(const char* name) { return std::getenv(name); }
#else
inline const char* dev_env(const char*) { return nullptr; }
#endif

}  // namespace crispy

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return ::crispy::fail(_e == hipErrorOutOfMemory ? CRISPY_ERR_OOM : CRISPY_ERR_HIP, "%s: %s", #expr, \
                            hipGetErrorString(_e));                                                \
  } while (0)

// Every extern "C" definition is a function-try-block closed by one of these: nothing unwinds into the caller
// (the reference host builds with panic=abort, Cargo.toml:10-20; a C++ exception crossing the FFI is UB there).
#define CRISPY_CATCH_RET(name) catch (...) { return ::crispy::fail_exception(name); }
#define CRISPY_CATCH_VOID(name) catch (...) { (void)::crispy::fail_exception(name); }
