// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of libcubehip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

#define CUBE_WAVE 64

// error reporting (thread-local text returned by cube_last_error)
void cube_set_error(const char* fmt, ...);
#define CUBE_FAIL(code, ...)      \
  do {                            \
    cube_set_error(__VA_ARGS__);  \
    return (code);                \
  } while (0)
// hipGetLastError() is per host thread and sticky: an error left behind by an unrelated call of the shared
// HIP runtime (a torch op, an earlier failed launch) must not be reported by the next cube_* entry
#define CUBE_ENTRY() ((void)hipGetLastError())
#define CUBE_CHECK_LAUNCH(name)                                                   \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) CUBE_FAIL(-3, "%s: launch failed: %s", name, hipGetErrorString(e__)); \
  } while (0)

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float x) {  // round-to-nearest-even (v_cvt_pk_bf16_f32)
  bf16_t h = (bf16_t)x;
  return __builtin_bit_cast(uint16_t, h);
}
// split an fp32 value into two bf16 terms: v ~= hi + lo with ~16 significant bits ("bf16x2")
__device__ __forceinline__ void split_bf16(float v, uint16_t& hi, uint16_t& lo) {
  hi = f32_to_bf16(v);
  lo = f32_to_bf16(v - bf16_to_f32(hi));
}
// v_exp_f32 + v_rcp_f32 forms (1 ulp each): absolute error < 3e-7 on (-1, 1) outputs, exact limits
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
  // tanh(x) = 1 - 2/(exp(2x)+1)
  const float e = __expf(2.0f * x);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// One LSTM cell (torch.nn.LSTM, gate order i, f, g, o) from the four pre-activations and the old cell state; shared by
// cube_lstm_cell and the CUBE_EPI_LSTM epilogue of the GEMM so that both give the same bits.  Explicit fma / mul: the two
// callers must not be contracted differently.
__device__ __forceinline__ void lstm_cell_update(float pi, float pf, float pg, float po, float c_old, float& c_new, float& h_new) {
  const float gi = sigmoidf_(pi), gf = sigmoidf_(pf), gg = tanhf_(pg), go = sigmoidf_(po);
  c_new = __fmaf_rn(gf, c_old, __fmul_rn(gi, gg));
  h_new = __fmul_rn(go, tanhf_(c_new));
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: remember per
// device (one process may drive several GPUs, from several host threads) what has been reserved for
// a kernel.  The table is atomic: two threads that race on the first launch both set the attribute
// (idempotent) and both record it; nothing else in the library is mutable process state.
constexpr int CUBE_MAX_DEVICES = 64;
struct CubeLdsReservation {
  std::atomic<size_t> bytes[CUBE_MAX_DEVICES] = {};
  // true if `lds` bytes are (now) reserved for `fn` on the current device
  bool ensure(const void* fn, size_t lds) {
    if (lds <= 65536) return true;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CUBE_MAX_DEVICES) return false;
    if (bytes[dev].load(std::memory_order_acquire) >= lds) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
    size_t seen = bytes[dev].load(std::memory_order_relaxed);
    while (seen < lds && !bytes[dev].compare_exchange_weak(seen, lds, std::memory_order_release)) {}
    return true;
  }
};
