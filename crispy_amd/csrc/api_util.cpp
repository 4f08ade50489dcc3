// api_util.cpp -- thread-local error message and device checks shared by the C ABI.
#include "api_util.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

namespace crispy {

namespace {
thread_local std::string g_last_error;
}

int fail(int code, const char* fmt, ...) {
  char buf[768];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

const char* last_error_cstr() { return g_last_error.c_str(); }

bool device_is_gfx950(int dev) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
  return std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

int check_device(int device, const char* who) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(CRISPY_ERR_NO_DEVICE, "%s: no HIP device (this library has no CPU path)", who);
  if (device < 0 || device >= ndev)
    return fail(CRISPY_ERR_INVALID_ARG, "%s: device %d out of range [0,%d)", who, device, ndev);
  if (!device_is_gfx950(device)) return fail(CRISPY_ERR_NO_DEVICE, "%s: device %d is not gfx950 (MI355X)", who, device);
  return CRISPY_OK;
}

}  // namespace crispy
