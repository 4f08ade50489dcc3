// api_util.h -- error plumbing shared by the extern "C" translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/crispy_hip.h"

namespace crispy {

// Sets the calling thread's last-error message and returns `code`.
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
const char* last_error_cstr();
// CRISPY_OK if `device` is a usable gfx950 device, else an error with `who` in the message.
int check_device(int device, const char* who);
bool device_is_gfx950(int dev);

}  // namespace crispy

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return ::crispy::fail(_e == hipErrorOutOfMemory ? CRISPY_ERR_OOM : CRISPY_ERR_HIP, "%s: %s", #expr, \
                            hipGetErrorString(_e));                                                \
  } while (0)
