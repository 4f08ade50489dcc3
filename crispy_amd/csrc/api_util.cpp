// api_util.cpp -- thread-local error message and device checks shared by the C ABI.
#include "api_util.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <exception>
#include <new>
#include <stdexcept>
#include <string>

namespace crispy {

namespace {
// fixed storage: recording an error must not allocate (it is what runs after a std::bad_alloc)
thread_local char g_last_error[768] = "";
}

int fail(int code, const char* fmt, ...) {
  char buf[sizeof(g_last_error)];      // fmt's arguments may point into g_last_error ("%s" of the previous message)
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  std::memcpy(g_last_error, buf, sizeof(buf));
  return code;
}

const char* last_error_cstr() { return g_last_error; }

int fail_exception(const char* where) noexcept {
  try {
    throw;
  } catch (const std::bad_alloc&) {
    return fail(CRISPY_ERR_OOM, "%s: host allocation failed (std::bad_alloc)", where);
  } catch (const std::length_error& e) {
    return fail(CRISPY_ERR_OOM, "%s: %s (std::length_error)", where, e.what());
  } catch (const std::exception& e) {
    return fail(CRISPY_ERR_HIP, "%s: unexpected C++ exception: %s", where, e.what());
  } catch (...) {
    return fail(CRISPY_ERR_HIP, "%s: unexpected non-standard C++ exception", where);
  }
}

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
