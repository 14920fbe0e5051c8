// tests/hostsim/fakehip/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
// The dozen HIP runtime calls the aligner driver (bowtie2_amd/csrc/bt2g_search.cpp) makes, on host memory, so that the driver itself --
// reader / device-stage / writer threads, batch ordering, --shard, mixed paired + unpaired input -- can be run in a container without a
// GPU behind tests/hostsim/driver_twin.cpp.  Never on the include path of the product build.
#ifndef BT2G_FAKE_HIP_RUNTIME_H_
#define BT2G_FAKE_HIP_RUNTIME_H_
#include <cstdlib>
#include <cstring>

typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };
enum { hipHostMallocPortable = 1 };
inline const char* hipGetErrorString(hipError_t) { return "fake HIP error"; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 2; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 2; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memcpy(d, s, n); return hipSuccess; }
#endif
