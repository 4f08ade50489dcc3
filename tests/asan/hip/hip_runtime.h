// hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY: a host-memory stand-in for the HIP runtime, found before the real header
// when tests/asan/harness.cpp is compiled (-I tests/asan), so that the library's device-free HOST logic -- the GGML reader
// and de-quantiser, the rnnoise-nu parser, whisper_full's decision logic, prompt building -- runs on the CPU box under
// AddressSanitizer / UndefinedBehaviorSanitizer (the library parses untrusted model files inside a
// panic = "abort" host, src-tauri/Cargo.toml:10-20).  "Device" memory is malloc'd host memory, so every copy the loaders
// make into it is bounds-checked; kernels do not exist (tests/asan/harness.cpp stubs the launchers).  Never part of the
// product build: crispy_amd/csrc/Makefile does not know this directory.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cmath>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))

typedef enum hipError_t { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101, hipErrorUnknown = 999 } hipError_t;
typedef struct shim_stream* hipStream_t;
typedef struct shim_event* hipEvent_t;
typedef struct shim_graph* hipGraph_t;
typedef struct shim_graph_exec* hipGraphExec_t;
typedef enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 } hipMemcpyKind;
typedef enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 } hipStreamCaptureMode;
typedef enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 } hipDeviceAttribute_t;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostRegisterDefault = 0 };
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; int multiProcessorCount; size_t totalGlobalMem; };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };

// a cap on one allocation: a corrupt header that asks for terabytes must come back as an out-of-memory status, not take the box down
static const size_t SHIM_MAX_ALLOC = (size_t)3 << 30;

template <class T> static inline hipError_t hipMalloc(T** p, size_t n) {
  if (n > SHIM_MAX_ALLOC) { *p = nullptr; return hipErrorOutOfMemory; }
  *p = static_cast<T*>(malloc(n ? n : 1));
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t = nullptr) {
  for (size_t r = 0; r < h; ++r) memmove(static_cast<char*>(d) + r * dp, static_cast<const char*>(s) + r * sp, w);
  return hipSuccess;
}
static inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset2DAsync(void* d, size_t p, int v, size_t w, size_t h, hipStream_t = nullptr) {
  for (size_t r = 0; r < h; ++r) memset(static_cast<char*>(d) + r * p, v, w);
  return hipSuccess;
}
static inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  memset(p, 0, sizeof(*p));
  strcpy(p->name, "host shim"); strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-"); p->multiProcessorCount = 256; p->totalGlobalMem = (size_t)288 << 30;
  return hipSuccess;
}
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(malloc(1)); return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = reinterpret_cast<hipStream_t>(malloc(1)); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipSuccess; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = reinterpret_cast<hipGraph_t>(malloc(1)); return hipSuccess; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, void*, void*, size_t) { *e = reinterpret_cast<hipGraphExec_t>(malloc(1)); return hipSuccess; }
static inline hipError_t hipGraphDestroy(hipGraph_t g) { free(g); return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t e) { free(e); return hipSuccess; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(malloc(1)); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(malloc(1)); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
static inline hipError_t hipHostUnregister(void*) { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory (host shim)" : "error (host shim)"; }
