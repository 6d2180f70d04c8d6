// Shared device helpers of the gfx950 kernels (included by the .hip files only).
// Wavefront = 64 lanes everywhere; no CUDA-isms, no portability layer.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <type_traits>

#include "kernels.h"

namespace ju {

namespace {

using f16 = _Float16;
using bf16 = __bf16;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <typename T>
using Vec8 = T __attribute__((ext_vector_type(8)));
template <typename T>
using Vec4 = T __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma32(Vec8<f16> a, Vec8<f16> b, f32x16 c) {
	return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(Vec8<bf16> a, Vec8<bf16> b, f32x16 c) {
	return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ReLU on values already rounded to the 16-bit type, as a packed signed-int16 max
// with 0: a negative bf16/f16 (sign bit set, -0.0 included) is a negative int16.
// Rounding keeps the sign, so this equals relu-then-round bit for bit, and it is 2
// v_pk_max_i16 per 4 values instead of 8 v_max_f32 (hipcc puts a canonicalising
// max in front of every fmaxf on an MFMA result).
template <typename T>
__device__ __forceinline__ Vec4<T> reluPacked(Vec4<T> v) {
	typedef short s16x4 __attribute__((ext_vector_type(4)));
	s16x4 b = __builtin_bit_cast(s16x4, v);
	const s16x4 z = {0, 0, 0, 0};
	b = __builtin_elementwise_max(b, z);
	return __builtin_bit_cast(Vec4<T>, b);
}

// tanh(x) = 1 - 2 / (exp(2x) + 1) on the hardware exp2 / rcp units (1 ulp each):
// absolute error < 4e-7, far below the 1/255 output step; saturates correctly
// (exp -> inf: rcp -> 0; exp -> 0: 1 - 2).  libm's tanhf is ~40 VALU ops per value
// and the tail evaluates 6.2 M of them per frame.
__device__ __forceinline__ float fastTanh(float x) {
	const float t = __builtin_amdgcn_exp2f(x * 2.885390081777927f);  // 2 * log2(e)
	return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}

// Opt a kernel in to more than 64 KiB of dynamic LDS.  The attribute is per DEVICE:
// a process may hold runtimes on several GPUs, so "done" is tracked per device (one
// mask per kernel instantiation, passed in by the launcher).  The first launch on a
// device happens in the engine's constructor, before any graph capture.
inline void ensureDynamicLds(const void *kern, int bytes, std::atomic<std::uint64_t> *doneMask,
    const char *what) {
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
	const std::uint64_t bit = 1ull << dev;
	if (doneMask->load(std::memory_order_acquire) & bit) return;
	const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
	if (e != hipSuccess) {
		throw std::runtime_error(std::string("hipFuncSetAttribute(") + what + " LDS): " + hipGetErrorString(e));
	}
	doneMask->fetch_or(bit, std::memory_order_release);
}

// four f32 -> packed 16-bit (RNE).  f16: a pairwise vector convert maps to one
// v_cvt_pk_f16_f32 per two values (element-wise casts cost a convert per value plus
// merges).  bf16: hipcc scalarises the vector form (more ops, not fewer), and an
// inline-asm v_cvt_pk_bf16_f32 is not padded with the MFMA -> VALU wait states, so the
// element-wise casts stay.
template <typename T>
__device__ __forceinline__ Vec4<T> pack4(float a, float b, float c, float d) {
	if constexpr (std::is_same<T, _Float16>::value) {
		typedef float f32x2p __attribute__((ext_vector_type(2)));
		typedef T t2p __attribute__((ext_vector_type(2)));
		const f32x2p lo = {a, b}, hi = {c, d};
		return __builtin_shufflevector(__builtin_convertvector(lo, t2p), __builtin_convertvector(hi, t2p),
		    0, 1, 2, 3);
	} else {
		return Vec4<T>{static_cast<T>(a), static_cast<T>(b), static_cast<T>(c), static_cast<T>(d)};
	}
}

inline void hipCheckLaunch(const char *what) {
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) {
		throw std::runtime_error(std::string("HIP launch failed (") + what +
		                         "): " + hipGetErrorString(e));
	}
}

inline unsigned blocksFor(size_t n) { return static_cast<unsigned>((n + 255) / 256); }

}  // namespace

}  // namespace ju
