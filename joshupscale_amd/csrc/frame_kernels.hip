// gfx950 (CDNA4, MI355X): the frame-level kernels around the convolutions.
//
//  * pack_frames        u8 BGRX frame + frame history -> 16-channel flow input
//  * frame_sums         normalize_brightness scalar
//  * maxpool2/upsample2 flow auto-encoder resampling (TF1 asymmetric bilinear)
//  * warp_pack          dense bilinear warp of the previous HR output fused with
//                       space-to-depth(4), the concat with the LR frame and the
//                       16-bit pack (reference models.py:799-801, 523-530)
//  * tail_fused         ConvT1 + ConvT2 on the matrix cores, bias, tanh, bilinear x4
//                       skip, clip, HR state write and truncating BGRX u8 pack
//                       (reference models.py:552-593, keras_layers.py:211-230,
//                       core/src/cuda_convert.cc.cu:95-108); tail = two-kernel form
//  * temporal_*         moving-average output filter (frame_moving_avg.py)
//  * copy_rows/to_float staging and introspection helpers
#include <algorithm>

#include "kernel_common.h"

namespace ju {

namespace {

// ---------------------------------------------------------------------------
// flow input packing
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void frame_sums_kernel(const std::uint8_t *__restrict__ frame,
    std::ptrdiff_t frameStride, int H, int W, unsigned *__restrict__ sums) {
	__shared__ unsigned part[3][16];
	unsigned s0 = 0, s1 = 0, s2 = 0;
	for (int i = threadIdx.x; i < H * W; i += 1024) {
		const int y = i / W, x = i - y * W;
		const unsigned v = *reinterpret_cast<const unsigned *>(frame + y * frameStride + x * 4);
		s0 += v & 0xff;
		s1 += (v >> 8) & 0xff;
		s2 += (v >> 16) & 0xff;
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		s0 += __shfl_down(s0, o);
		s1 += __shfl_down(s1, o);
		s2 += __shfl_down(s2, o);
	}
	const int wv = threadIdx.x >> 6;
	if ((threadIdx.x & 63) == 0) {
		part[0][wv] = s0;
		part[1][wv] = s1;
		part[2][wv] = s2;
	}
	__syncthreads();
	// (a thread's and a wave's sum fit 32 bits for every frame the loader admits -- 67 M pixels / 1024 threads x 255 x 64;
	// the frame's does not beyond 16.8 M pixels: 64 bits, stored low word first.  Until round 4 it was one 32-bit word.)
	if (threadIdx.x < 3) {
		unsigned long long t = 0;
		for (int k = 0; k < 16; ++k) t += part[threadIdx.x][k];
		sums[2 * threadIdx.x] = static_cast<unsigned>(t);
		sums[2 * threadIdx.x + 1] = static_cast<unsigned>(t >> 32);
	}
}

// packed flow input [N][16] -> the first two 16-byte chunks of 64-channel records
// [N][64] (the other chunks stay zero from allocation): input of the flow-resnet's
// resident tower, whose layer 0 reads 64-channel pixels.
__global__ __launch_bounds__(256) void expand_channels_kernel(const uint4 *__restrict__ in,
    uint4 *__restrict__ out, int nPix) {
	const int i = blockIdx.x * 256 + threadIdx.x;  // one 16-byte chunk
	if (i >= nPix * 2) return;
	out[(size_t)(i >> 1) * 8 + (i & 1)] = in[i];
}

template <typename T>
__global__ __launch_bounds__(256) void pack_frames_kernel(const std::uint8_t *__restrict__ frame,
    std::ptrdiff_t frameStride, const T *__restrict__ prev, T *__restrict__ cur, int H, int W,
    int PH, int PW, int padTop, int padLeft, int numInputs, const unsigned *__restrict__ sums) {
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= PH * PW) return;
	const float bright = brightnessOf(sums, 1.0f / static_cast<float>(H * W));
	const int py = idx / PW;
	const int pxx = idx - py * PW;
	const int y = py - padTop;
	const int x = pxx - padLeft;
	float c0 = 0.f, c1 = 0.f, c2 = 0.f;  // ZeroPadding2D after preprocess: 0.0 in the border
	if (y >= 0 && y < H && x >= 0 && x < W) {
		const unsigned v = *reinterpret_cast<const unsigned *>(frame + y * frameStride + x * 4);
		// the flow net sees the brightness-normalised frame (models.py:779); the pad
		// border stays exactly zero (ZeroPadding2D comes after the subtraction)
		c0 = preprocessU8(v & 0xff) - bright;
		c1 = preprocessU8((v >> 8) & 0xff) - bright;
		c2 = preprocessU8((v >> 16) & 0xff) - bright;
	}
	const Vec8<T> p0 = *reinterpret_cast<const Vec8<T> *>(prev + (size_t)idx * 16);
	const Vec8<T> p1 = *reinterpret_cast<const Vec8<T> *>(prev + (size_t)idx * 16 + 8);
	T pv[16];
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		pv[i] = p0[i];
		pv[8 + i] = p1[i];
	}
	const int nch = 3 * numInputs;
	T o[16];
	o[0] = static_cast<T>(c0);
	o[1] = static_cast<T>(c1);
	o[2] = static_cast<T>(c2);
#pragma unroll
	for (int k = 3; k < 16; ++k) o[k] = (k < nch) ? pv[k - 3] : static_cast<T>(0.f);
	Vec8<T> o0, o1;
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		o0[i] = o[i];
		o1[i] = o[8 + i];
	}
	*reinterpret_cast<Vec8<T> *>(cur + (size_t)idx * 16) = o0;
	*reinterpret_cast<Vec8<T> *>(cur + (size_t)idx * 16 + 8) = o1;
}

// ---------------------------------------------------------------------------
// 2x2 max-pool and TF1 bilinear x2 (8 channels = 16 B per thread)
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_kernel(
    const T *__restrict__ in, T *__restrict__ out, int H, int W, int C) {
	const int OH = H / 2, OW = W / 2, CC = C / 8;
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= OH * OW * CC) return;
	const int c = idx % CC;
	const int pix = idx / CC;
	const int ox = pix % OW;
	const int oy = pix / OW;
	const T *base = in + ((size_t)(2 * oy) * W + 2 * ox) * C + c * 8;
	const Vec8<T> a = *reinterpret_cast<const Vec8<T> *>(base);
	const Vec8<T> b = *reinterpret_cast<const Vec8<T> *>(base + C);
	const Vec8<T> d = *reinterpret_cast<const Vec8<T> *>(base + (size_t)W * C);
	const Vec8<T> e = *reinterpret_cast<const Vec8<T> *>(base + (size_t)W * C + C);
	Vec8<T> o;
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		const float m = fmaxf(fmaxf(static_cast<float>(a[i]), static_cast<float>(b[i])),
		    fmaxf(static_cast<float>(d[i]), static_cast<float>(e[i])));
		o[i] = static_cast<T>(m);
	}
	*reinterpret_cast<Vec8<T> *>(out + (size_t)pix * C + c * 8) = o;
}

template <typename T>
__global__ __launch_bounds__(256) void upsample2_kernel(
    const T *__restrict__ in, T *__restrict__ out, int H, int W, int C) {
	// (blockIdx.y: the frame of a look-ahead launch, dense tensors one after the other)
	in += (size_t)blockIdx.y * H * W * C;
	out += (size_t)blockIdx.y * H * W * C * 4;
	// tf.compat.v1.image.resize_bilinear(align_corners=False,
	// half_pixel_centers=False): src = dst / 2 (reference keras_layers.py:46-52)
	const int OH = H * 2, OW = W * 2, CC = C / 8;
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= OH * OW * CC) return;
	const int c = idx % CC;
	const int pix = idx / CC;
	const int ox = pix % OW;
	const int oy = pix / OW;
	const int y0 = oy >> 1, x0 = ox >> 1;
	const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
	const float fy = (oy & 1) * 0.5f, fx = (ox & 1) * 0.5f;
	const Vec8<T> tl = *reinterpret_cast<const Vec8<T> *>(in + ((size_t)y0 * W + x0) * C + c * 8);
	const Vec8<T> tr = *reinterpret_cast<const Vec8<T> *>(in + ((size_t)y0 * W + x1) * C + c * 8);
	const Vec8<T> bl = *reinterpret_cast<const Vec8<T> *>(in + ((size_t)y1 * W + x0) * C + c * 8);
	const Vec8<T> br = *reinterpret_cast<const Vec8<T> *>(in + ((size_t)y1 * W + x1) * C + c * 8);
	Vec8<T> o;
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		const float a = static_cast<float>(tl[i]), b = static_cast<float>(tr[i]);
		const float d = static_cast<float>(bl[i]), e = static_cast<float>(br[i]);
		const float top = a + (b - a) * fx;
		const float bot = d + (e - d) * fx;
		o[i] = static_cast<T>(top + (bot - top) * fy);
	}
	*reinterpret_cast<Vec8<T> *>(out + (size_t)pix * C + c * 8) = o;
}

// ---------------------------------------------------------------------------
// dense warp + space-to-depth + concat + pack
// ---------------------------------------------------------------------------
// One thread per (LR pixel, HR row i of its 4x4 block): 4 warped HR pixels x 3
// channels plus 4 spare slots = one 32-byte quarter of the pixel's 128-byte
// generator-input record.  Four consecutive lanes fill one record, a wavefront
// writes 2 KiB contiguously.
template <typename T>
__global__ __launch_bounds__(256) void warp_pack_kernel(const f16 *__restrict__ state,
    const f16 *__restrict__ flow, const std::uint8_t *__restrict__ frame,
    std::ptrdiff_t frameStride, T *__restrict__ out, int outPitch, int H, int W, int PW, int padTop,
    int padLeft, const unsigned *__restrict__ sums, f16 *__restrict__ preWarpOut) {
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= H * W * 4) return;
	const float bright = brightnessOf(sums, 1.0f / static_cast<float>(H * W));  // pre_warp += b (models.py:803)
	const int i = idx & 3;
	const int pix = idx >> 2;
	const int w = pix % W;
	const int h = pix / W;
	// depth-to-space(4) of the flow head is just this channel addressing:
	// flow[4h+i, 4w+j, k] = head[h, w, (i*4+j)*2 + k]  (keras_layers.py:175)
	const Vec8<f16> f8 = *reinterpret_cast<const Vec8<f16> *>(
	    flow + ((size_t)(h + padTop) * PW + (w + padLeft)) * 32 + i * 8);
	Vec8<T> o0, o1;
	Vec4<f16> pw[4];
	warpQuarter<T>(state, f8, frame, frameStride, H, W, h, w, i, bright, o0, o1, pw);
	if (preWarpOut != nullptr) {
		f16 *d = preWarpOut + ((size_t)(4 * h + i) * (4 * W) + 4 * w) * 4;
#pragma unroll
		for (int j = 0; j < 4; ++j) *reinterpret_cast<Vec4<f16> *>(d + 4 * j) = pw[j];
	}
	// (out is addressed at image pixel (0, 0) with a row pitch: the generator input lives in the tower layout,
	// whose zero border is the first convolution's padding and is never written)
	T *dst = out + ((size_t)h * outPitch + w) * 64 + i * 16;
	*reinterpret_cast<Vec8<T> *>(dst) = o0;
	*reinterpret_cast<Vec8<T> *>(dst + 8) = o1;
}

// ---------------------------------------------------------------------------
// temporal moving-average output filter (scripts/inference/onnx/frame_moving_avg.py
// :146-302, its default mode: global L1 scene-cut gate with a sign function)
// ---------------------------------------------------------------------------
// Runs after the tail: `state` holds gen - b (f16, b = brightness scalar or 0),
// `preWarp` the warped previous output (+ b).  Pass 1 sums |gen - pre_warp| over
// every element into a 32.32 fixed-point accumulator (integer atomics: the result
// does not depend on the order, so the gate is deterministic); pass 2 blends and
// rewrites the state and the u8 frame unless the gate says "scene cut" (then the
// generator output written by the tail already is the result).
constexpr double kTemporalScale = 4294967296.0;  // 2^32

struct TemporalGeom {
	int HH, WW;          // HR frame
	int window;          // 0: global
	int GH, GW;          // gate grid (1 x 1 when global)
	int padY, padX;      // leading zero padding of the windowed mean (:208-215)
	float strength, threshold, gain;
	int l2, limit, luma;
	int lrPixels;
};

// Clears the gate accumulators.  A kernel, not hipMemsetAsync: the per-frame program is
// replayed from a captured hipGraph, and the memset NODE of a 768-byte clear was observed
// to leave words uncleared on replay under the HIP 7.0 runtime that PyTorch bundles (the
// process uses that runtime whenever torch is imported first, as bench.py and the tests
// do); eager launches and the ROCm 7.2 runtime were fine.  A kernel node has no such mode.
__global__ __launch_bounds__(256) void zero_words_kernel(unsigned long long *__restrict__ p, int n) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i < n) p[i] = 0ull;
}

// one element's contribution to the gate statistic (:157-187, 219-222)
__device__ __forceinline__ float temporalTerm(float gen, float pw, int ch, const TemporalGeom &g) {
	if (g.limit) pw = fmaxf(fminf(pw, 0.5f), -0.5f);
	float d = gen - pw;
	d = g.l2 ? d * d : fabsf(d);
	if (g.luma) {
		const float k = (ch == 0 ? 0.1140f : (ch == 1 ? 0.5870f : 0.2989f)) * 3.0f;  // LUMA_NORM (:95-96)
		d *= g.l2 ? k * k : k;
	}
	return d;
}

__global__ __launch_bounds__(256) void temporal_reduce_kernel(const f16 *__restrict__ state,
    const f16 *__restrict__ preWarp, TemporalGeom g, const unsigned *__restrict__ sums,
    unsigned long long *__restrict__ acc) {
	const float bright = brightnessOf(sums, 1.0f / static_cast<float>(g.lrPixels));
	const size_t nPix = (size_t)g.HH * g.WW;
	if (g.window == 0) {
		float s = 0.f;
		for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nPix; i += (size_t)gridDim.x * 256) {
			const Vec4<f16> gv = *reinterpret_cast<const Vec4<f16> *>(state + i * 4);
			const Vec4<f16> q = *reinterpret_cast<const Vec4<f16> *>(preWarp + i * 4);
#pragma unroll
			for (int c = 0; c < 3; ++c) {
				s += temporalTerm(static_cast<float>(gv[c]) + bright, static_cast<float>(q[c]), c, g);
			}
		}
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
		if ((threadIdx.x & 63) == 0) {
			atomicAdd(acc, static_cast<unsigned long long>(static_cast<double>(s) * kTemporalScale + 0.5));
		}
		return;
	}
	// windowed: one thread per HR pixel; integer (fixed-point) atomics per window, so the
	// sums do not depend on the order.  A wave whose 64 pixels fall into one window
	// (the common case: rows are contiguous) reduces first and issues one atomic.
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	float s = 0.f;
	int widx = -1;
	if (i < nPix) {
		const int y = static_cast<int>(i / g.WW), x = static_cast<int>(i - (size_t)y * g.WW);
		widx = ((y + g.padY) / g.window) * g.GW + (x + g.padX) / g.window;
		const Vec4<f16> gv = *reinterpret_cast<const Vec4<f16> *>(state + i * 4);
		const Vec4<f16> q = *reinterpret_cast<const Vec4<f16> *>(preWarp + i * 4);
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			s += temporalTerm(static_cast<float>(gv[c]) + bright, static_cast<float>(q[c]), c, g);
		}
	}
	unsigned long long fx = static_cast<unsigned long long>(static_cast<double>(s) * kTemporalScale + 0.5);
	const int first = __shfl(widx, 0);
	if (__all(widx == first)) {
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) fx += __shfl_xor(fx, off);
		if ((threadIdx.x & 63) == 0 && first >= 0) atomicAdd(acc + first, fx);
	} else if (widx >= 0) {
		atomicAdd(acc + widx, fx);
	}
}

// gate value of one grid cell: sign(m - t) or tanh(gain * (m - t)) (:223-238)
__device__ __forceinline__ float temporalGate(const unsigned long long *acc, int cell, double denom,
    const TemporalGeom &g) {
	const double mean = static_cast<double>(acc[cell]) / kTemporalScale / denom;
	const double d = mean - static_cast<double>(g.threshold);
	if (g.gain == 0.0f) return d > 0.0 ? 1.0f : (d < 0.0 ? -1.0f : 0.0f);
	return tanhf(static_cast<float>(d) * g.gain);
}

__global__ __launch_bounds__(256) void temporal_blend_kernel(f16 *__restrict__ state,
    const f16 *__restrict__ preWarp, std::uint8_t *__restrict__ outU8, std::ptrdiff_t outStride,
    TemporalGeom g, const unsigned *__restrict__ sums, const unsigned long long *__restrict__ acc) {
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= (size_t)g.HH * g.WW) return;
	const int y = static_cast<int>(i / g.WW), x = static_cast<int>(i - (size_t)y * g.WW);
	float c;
	if (g.window == 0) {
		c = temporalGate(acc, 0, 3.0 * g.HH * g.WW, g);
		// hard gate, scene cut: out = gen exactly, which the tail already wrote (with
		// --limit too: the weight of pre_warp is 0)
		if (g.gain == 0.0f && c > 0.0f) return;
	} else {
		// Resize(linear, asymmetric) of the gate grid by `window`, then the un-pad slice
		// (:239-270): src = dst / window, lo = floor, hi = min(lo + 1, n - 1)
		const double denom = 3.0 * g.window * g.window;
		const float sy = static_cast<float>(y + g.padY) / static_cast<float>(g.window);
		const float sx = static_cast<float>(x + g.padX) / static_cast<float>(g.window);
		const int y0 = static_cast<int>(sy), x0 = static_cast<int>(sx);
		const int y1 = min(y0 + 1, g.GH - 1), x1 = min(x0 + 1, g.GW - 1);
		const float fy = sy - static_cast<float>(y0), fx = sx - static_cast<float>(x0);
		const float c00 = temporalGate(acc, y0 * g.GW + x0, denom, g), c01 = temporalGate(acc, y0 * g.GW + x1, denom, g);
		const float c10 = temporalGate(acc, y1 * g.GW + x0, denom, g), c11 = temporalGate(acc, y1 * g.GW + x1, denom, g);
		const float top = c00 + (c01 - c00) * fx, bot = c10 + (c11 - c10) * fx;
		c = top + (bot - top) * fy;
	}
	const float half = 0.5f * g.strength;
	const float m1 = half - c * half;         // weight of pre_warp (:272-279)
	const float m2 = c * half + 1.0f - half;  // weight of the generator output (:280-285)
	const float bright = brightnessOf(sums, 1.0f / static_cast<float>(g.lrPixels));
	const Vec4<f16> gv = *reinterpret_cast<const Vec4<f16> *>(state + i * 4);
	const Vec4<f16> q = *reinterpret_cast<const Vec4<f16> *>(preWarp + i * 4);
	Vec4<f16> st;
	unsigned packed = 0;
#pragma unroll
	for (int ch = 0; ch < 3; ++ch) {
		float pw = static_cast<float>(q[ch]);
		if (g.limit) pw = fmaxf(fminf(pw, 0.5f), -0.5f);
		const float r = pw * m1 + (static_cast<float>(gv[ch]) + bright) * m2;
		st[ch] = static_cast<f16>(r - bright);
		const unsigned u = static_cast<unsigned>((r + 0.5f) * 255.0f);  // postprocess, truncating
		packed |= (u & 0xff) << (8 * ch);
	}
	st[3] = static_cast<f16>(0.f);
	*reinterpret_cast<Vec4<f16> *>(state + i * 4) = st;
	*reinterpret_cast<unsigned *>(outU8 + y * outStride + (std::ptrdiff_t)x * 4) = packed;
}

// ---------------------------------------------------------------------------
// generator tail
// ---------------------------------------------------------------------------
// One thread per mid-resolution pixel (2h+a, 2w+b): 32 channels in, 2x2 HR
// pixels x 3 channels out.
template <typename T>
__global__ __launch_bounds__(256) void tail_kernel(const T *__restrict__ y,
    const float *__restrict__ w2, const float *__restrict__ b2,
    const std::uint8_t *__restrict__ frame, std::ptrdiff_t frameStride,
    f16 *__restrict__ stateOut, std::uint8_t *__restrict__ outU8, std::ptrdiff_t outStride,
    int H, int W, const unsigned *__restrict__ sums) {
	const float bright = brightnessOf(sums, 1.0f / static_cast<float>(H * W));
	const int MW = 2 * W, MH = 2 * H;
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= MW * MH) return;
	const int mx = idx % MW;
	const int my = idx / MW;
	const int h = my >> 1, a = my & 1;
	const int w = mx >> 1, b = mx & 1;
	const T *src = y + ((size_t)h * W + w) * 128 + (a * 2 + b) * 32;
	float in[32];
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const Vec8<T> v = *reinterpret_cast<const Vec8<T> *>(src + k * 8);
#pragma unroll
		for (int i = 0; i < 8; ++i) in[k * 8 + i] = static_cast<float>(v[i]);
	}
	// LR neighbourhood for the bilinear x4 skip (UpscaleLayer, keras_layers.py:46-52)
	const int h1 = min(h + 1, H - 1), w1 = min(w + 1, W - 1);
	float lr[2][2][3];
#pragma unroll
	for (int yy = 0; yy < 2; ++yy) {
#pragma unroll
		for (int xx = 0; xx < 2; ++xx) {
			const unsigned v = *reinterpret_cast<const unsigned *>(
			    frame + (yy ? h1 : h) * frameStride + (xx ? w1 : w) * 4);
			lr[yy][xx][0] = preprocessU8(v & 0xff);
			lr[yy][xx][1] = preprocessU8((v >> 8) & 0xff);
			lr[yy][xx][2] = preprocessU8((v >> 16) & 0xff);
		}
	}
	const int WW = 4 * W;
#pragma unroll
	for (int a2 = 0; a2 < 2; ++a2) {
		const int Y = 2 * my + a2;
		const float fy = static_cast<float>(Y & 3) * 0.25f;
		Vec8<f16> st;
		unsigned pk[2];
#pragma unroll
		for (int b2i = 0; b2i < 2; ++b2i) {
			const int X = 2 * mx + b2i;
			const float fx = static_cast<float>(X & 3) * 0.25f;
			unsigned packed = 0;
#pragma unroll
			for (int c = 0; c < 3; ++c) {
				// ConvT 2x2 s2: y[2h+a,2w+b,o] = sum_c x[h,w,c] K[a,b,o,c] (+bias)
				float acc = b2[c];
				const float *wk = w2 + ((a2 * 2 + b2i) * 3 + c) * 32;
#pragma unroll
				for (int k = 0; k < 32; ++k) acc = fmaf(in[k], wk[k], acc);
				const float top = lr[0][0][c] + (lr[0][1][c] - lr[0][0][c]) * fx;
				const float bot = lr[1][0][c] + (lr[1][1][c] - lr[1][0][c]) * fx;
				const float skip = top + (bot - top) * fy;
				float r = fastTanh(acc) + skip;
				r = fminf(fmaxf(r, -0.5f), 0.5f);  // ClipLayer
				st[b2i * 4 + c] = static_cast<f16>(r - bright);  // fed-back state: output_raw - b (models.py:810)
				// PostprocessLayer + truncating cast (cuda_convert.cc.cu:76-81)
				const unsigned u = static_cast<unsigned>((r + 0.5f) * 255.0f);
				packed |= (u & 0xff) << (8 * c);
			}
			st[b2i * 4 + 3] = static_cast<f16>(0.f);
			pk[b2i] = packed;  // X byte = 0
		}
		*reinterpret_cast<Vec8<f16> *>(stateOut + ((size_t)Y * WW + 2 * mx) * 4) = st;
		*reinterpret_cast<uint2 *>(outU8 + Y * outStride + 2 * mx * 4) = make_uint2(pk[0], pk[1]);
	}
}


// ---------------------------------------------------------------------------
// fused generator tail on the matrix cores
// ---------------------------------------------------------------------------
// trunk [H][W][64] -> ConvT(2x2,s2,64->32)+BN+ReLU -> ConvT(2x2,s2,32->3)+bias -> tanh
// -> + bilinear x4 of the LR frame -> clip -> HR state (f16) and BGRX u8, in ONE
// pass (reference models.py:559-593, keras_layers.py:211-230, cuda_convert.cc.cu:76-81).
// The two-kernel form wrote and re-read a [H][W][128] tensor (66 MB per frame).
//   stage 1  D1[128][32 px] = W1[128][64] x X[64][32 px]: 4 cout blocks x 4 k-steps.
//            Cout block nb = (a*2+b) IS the mid-resolution pixel (2h+a, 2w+b)'s 32
//            channels, so after bias + ReLU each block goes to LDS pixel-major
//            (64 B per mid pixel) and is directly the B operand of
//   stage 2  D2[16][32 mid px] = W2[16][32] x Y[32][32 mid px], rows m = 4*(a'*2+b') + c:
//            lane (mid px, hh), register group g2 then holds the 3 channels of ONE HR
//            pixel (a' = g2, b' = hh).
// One workgroup = 8 LR rows x 32 px (4 waves x 2 rows); outputs are staged in LDS
// and written as whole rows, 16 B per lane.
// LDS (72 KiB, two workgroups per CU): per wave ONE input row (4 KiB; the f16 state
// rows of the output staging reuse it once stage 1 has consumed it), 16 KiB convT1
// weights, per wave 8 KiB of mid pixels and 2 KiB of u8 output staging.

struct TailFusedParams {
	const void *x;        // trunk, addressed at image pixel (0,0)
	int xPitch;           // row pitch in pixels
	const void *w1;       // convT1 as 1x1 conv 64->128, packConvWeights order with nb = 2
	const float *b1;      // [128]
	const void *w2;       // A fragments of convT2: [2 ks][64 lanes][8] 16-bit
	const float *b2;      // [3]
	const std::uint8_t *frame;
	std::ptrdiff_t frameStride;
	void *state;          // f16 [4H][4W][4]
	std::uint8_t *outU8;
	std::ptrdiff_t outStride;
	const unsigned *sums;
	int H, W;
	float slope;          // < 0: ReLU after convT1, else LeakyReLU(slope)
};

template <typename T>
__global__ __launch_bounds__(256) void tail_fused_kernel(TailFusedParams p) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, px = lane & 31, hh = lane >> 5;
	unsigned char *smRow = smem + wave * kTailLdsRow;  // this wave's current input row / state staging
	unsigned char *smW = smem + kTailLdsIn;
	unsigned char *smMid = smem + kTailLdsIn + kTailLdsW1 + wave * (kTailLdsMid / 4);
	unsigned char *smU8 = smem + kTailLdsIn + kTailLdsW1 + kTailLdsMid + wave * (kTailLdsU8 / 4);
	const int tx0 = blockIdx.x * 32, ty0 = blockIdx.y * 8;
	const T *__restrict__ x = static_cast<const T *>(p.x);
	const float bright = brightnessOf(p.sums, 1.0f / static_cast<float>(p.H * p.W));

	// ---- weights (linear) for the workgroup; each wave fetches its own rows: the row
	//      after the current one travels in registers while the current one is processed ----
	// (four named registers and macros: an array that lives across the wave barriers
	// below, or a lambda capturing it, is left in scratch memory by hipcc)
#define JU_TAIL_ROW_ADDR(LR, K)                                                                  \
	(x + ((size_t)min(ty0 + (LR), p.H - 1) * p.xPitch + min(tx0 + ((lane + (K) * 64) >> 3), p.W - 1)) * 64 + \
	    ((lane + (K) * 64) & 7) * 8)
#define JU_TAIL_LOAD_ROW(LR)                                                    \
	row0 = *reinterpret_cast<const uint4 *>(JU_TAIL_ROW_ADDR(LR, 0));           \
	row1 = *reinterpret_cast<const uint4 *>(JU_TAIL_ROW_ADDR(LR, 1));           \
	row2 = *reinterpret_cast<const uint4 *>(JU_TAIL_ROW_ADDR(LR, 2));           \
	row3 = *reinterpret_cast<const uint4 *>(JU_TAIL_ROW_ADDR(LR, 3));
#define JU_TAIL_ROW_LDS(K)                                                                      \
	(smRow + ((lane + (K) * 64) >> 3) * 128 +                                                   \
	    ((((lane + (K) * 64) & 7) ^ ((((lane + (K) * 64) >> 3) >> 1) & 7)) << 4))
	uint4 row0, row1, row2, row3;
	JU_TAIL_LOAD_ROW(wave * 2)
	{
		const uint4 *src = reinterpret_cast<const uint4 *>(p.w1);
		uint4 *dst = reinterpret_cast<uint4 *>(smW);
#pragma unroll
		for (int k = 0; k < kTailLdsW1 / 16 / 256; ++k) dst[tid + k * 256] = src[tid + k * 256];
	}
	// convT2 A fragments (2 k-steps) and biases in registers
	Vec8<T> a2[2];
	a2[0] = reinterpret_cast<const Vec8<T> *>(p.w2)[lane];
	a2[1] = reinterpret_cast<const Vec8<T> *>(p.w2)[64 + lane];
	const float b2v[3] = {p.b2[0], p.b2[1], p.b2[2]};
	const TailRowArgs args{p.b1, p.frame, p.frameStride, p.state, p.outU8, p.outStride, p.H, p.W, p.slope};
	__syncthreads();

#pragma unroll
	for (int rw = 0; rw < 2; ++rw) {
		const int lr = wave * 2 + rw;  // LR row inside the tile
		const int h = ty0 + lr;
		// this row -> LDS (the previous row's output staging was read out at the end of
		// the last iteration); the next row's loads go out before the MFMAs
		*reinterpret_cast<uint4 *>(JU_TAIL_ROW_LDS(0)) = row0;
		*reinterpret_cast<uint4 *>(JU_TAIL_ROW_LDS(1)) = row1;
		*reinterpret_cast<uint4 *>(JU_TAIL_ROW_LDS(2)) = row2;
		*reinterpret_cast<uint4 *>(JU_TAIL_ROW_LDS(3)) = row3;
		if (rw == 0) {
			JU_TAIL_LOAD_ROW(lr + 1)
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		const auto fetchB = [&](int ks) {
			const int c = ks * 2 + hh;
			return *reinterpret_cast<const Vec8<T> *>(smRow + px * 128 + ((c ^ ((px >> 1) & 7)) << 4));
		};
		tailRow<T>(fetchB, smW, smMid, smRow, smU8, a2, b2v, args, bright, tx0, h, lane);
	}
}

#undef JU_TAIL_LOAD_ROW
#undef JU_TAIL_ROW_ADDR
#undef JU_TAIL_ROW_LDS

template <typename T>
void launchTailFusedT(const TailFusedParams &p, hipStream_t stream) {
	auto kern = tail_fused_kernel<T>;
	static std::atomic<std::uint64_t> ldsDone{0};
	ensureDynamicLds(reinterpret_cast<const void *>(kern), kTailLds, &ldsDone, "tail");
	dim3 grid((p.W + 31) / 32, (p.H + 7) / 8);
	hipLaunchKernelGGL(kern, grid, dim3(256), kTailLds, stream, p);
	hipCheckLaunch("tail_fused");
}

// ---------------------------------------------------------------------------
// staging helpers
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy_rows_kernel(const std::uint8_t *__restrict__ src,
    std::ptrdiff_t srcStride, std::uint8_t *__restrict__ dst, std::ptrdiff_t dstStride,
    unsigned wordsPerRow, unsigned rows) {
	const unsigned idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= wordsPerRow * rows) return;
	const unsigned r = idx / wordsPerRow;
	const unsigned c = idx - r * wordsPerRow;
	const unsigned v = *reinterpret_cast<const unsigned *>(
	    src + static_cast<std::ptrdiff_t>(r) * srcStride + c * 4);
	*reinterpret_cast<unsigned *>(dst + static_cast<std::ptrdiff_t>(r) * dstStride + c * 4) = v;
}

// One more frame of a look-ahead pass is complete: a count in host-mapped memory the thread blocked in
// Engine::processBatch polls, so that it can copy that frame out while the next one runs (host frames inside passes).
// The kernel boundary in front of it has released the frame's bytes to memory; the increment itself is a system-scope
// release so that the host sees it at once.
__global__ void signal_host_kernel(unsigned *word) {
	__hip_atomic_fetch_add(word, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <typename T>
__global__ __launch_bounds__(256) void to_float_kernel(
    const T *__restrict__ in, float *__restrict__ out, size_t n) {
	const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (idx < n) out[idx] = static_cast<float>(in[idx]);
}

// max |x| of a 16-bit tensor -> atomic max into *out (bit pattern of a non-negative float:
// integer order = float order).  Calibration mode only (Engine, JU_CALIBRATE=1).
template <typename T>
__global__ __launch_bounds__(256) void abs_max_kernel(const T *__restrict__ in, size_t n8, unsigned *__restrict__ out) {
	float m = 0.f;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
		const Vec8<T> v = reinterpret_cast<const Vec8<T> *>(in)[i];
#pragma unroll
		for (int k = 0; k < 8; ++k) m = fmaxf(m, fabsf(static_cast<float>(v[k])));
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
	if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
}

}  // namespace

void launchAbsMax(DType dt, const void *in, std::size_t n, unsigned *out, hipStream_t stream) {
	const std::size_t n8 = n / 8;  // (every tensor of the engine is a multiple of 8 elements)
	const unsigned nb = static_cast<unsigned>(std::min<std::size_t>(blocksFor(n8), 2048));
	if (dt == kF16) hipLaunchKernelGGL(abs_max_kernel<f16>, dim3(nb), dim3(256), 0, stream, static_cast<const f16 *>(in), n8, out);
	else hipLaunchKernelGGL(abs_max_kernel<bf16>, dim3(nb), dim3(256), 0, stream, static_cast<const bf16 *>(in), n8, out);
	hipCheckLaunch("abs_max");
}

void launchPackFrames(DType dt, const std::uint8_t *frame, std::ptrdiff_t frameStride,
    const void *prevPacked, void *curPacked, int H, int W, int PH, int PW, int padTop,
    int padLeft, int numInputs, const unsigned *sums,
    hipStream_t stream) {
	const unsigned nb = blocksFor((size_t)PH * PW);
	if (dt == kF16) {
		hipLaunchKernelGGL(pack_frames_kernel<f16>, dim3(nb), dim3(256), 0, stream, frame,
		    frameStride, static_cast<const f16 *>(prevPacked), static_cast<f16 *>(curPacked), H, W,
		    PH, PW, padTop, padLeft, numInputs, sums);
	} else {
		hipLaunchKernelGGL(pack_frames_kernel<bf16>, dim3(nb), dim3(256), 0, stream, frame,
		    frameStride, static_cast<const bf16 *>(prevPacked), static_cast<bf16 *>(curPacked), H,
		    W, PH, PW, padTop, padLeft, numInputs, sums);
	}
	hipCheckLaunch("pack_frames");
}

void launchExpandChannels(DType, const void *in16, void *out64, int nPix, hipStream_t stream) {
	hipLaunchKernelGGL(expand_channels_kernel, dim3(blocksFor((size_t)nPix * 2)), dim3(256), 0, stream,
	    static_cast<const uint4 *>(in16), static_cast<uint4 *>(out64), nPix);
	hipCheckLaunch("expand_channels");
}

void launchMaxPool2(DType dt, const void *in, void *out, int H, int W, int C, hipStream_t stream) {
	const unsigned nb = blocksFor((size_t)(H / 2) * (W / 2) * (C / 8));
	if (dt == kF16) {
		hipLaunchKernelGGL(maxpool2_kernel<f16>, dim3(nb), dim3(256), 0, stream,
		    static_cast<const f16 *>(in), static_cast<f16 *>(out), H, W, C);
	} else {
		hipLaunchKernelGGL(maxpool2_kernel<bf16>, dim3(nb), dim3(256), 0, stream,
		    static_cast<const bf16 *>(in), static_cast<bf16 *>(out), H, W, C);
	}
	hipCheckLaunch("maxpool2");
}

void launchUpsample2(DType dt, const void *in, void *out, int H, int W, int C, hipStream_t stream, int items) {
	const unsigned nb = blocksFor((size_t)(H * 2) * (W * 2) * (C / 8));
	if (launchesAreDry()) return;
	if (dt == kF16) {
		hipLaunchKernelGGL(upsample2_kernel<f16>, dim3(nb, items > 1 ? items : 1), dim3(256), 0, stream,
		    static_cast<const f16 *>(in), static_cast<f16 *>(out), H, W, C);
	} else {
		hipLaunchKernelGGL(upsample2_kernel<bf16>, dim3(nb, items > 1 ? items : 1), dim3(256), 0, stream,
		    static_cast<const bf16 *>(in), static_cast<bf16 *>(out), H, W, C);
	}
	hipCheckLaunch("upsample2");
}

void launchWarpPack(DType dt, const void *state, const void *flow, const std::uint8_t *frame,
    std::ptrdiff_t frameStride, void *out, int outPitch, int H, int W, int PW, int padTop, int padLeft,
    const unsigned *sums, void *preWarpOut, hipStream_t stream) {
	if (outPitch <= 0) outPitch = W;
	const unsigned nb = blocksFor((size_t)H * W * 4);
	if (dt == kF16) {
		hipLaunchKernelGGL(warp_pack_kernel<f16>, dim3(nb), dim3(256), 0, stream,
		    static_cast<const f16 *>(state), static_cast<const f16 *>(flow), frame, frameStride, static_cast<f16 *>(out), outPitch, H,
		    W, PW, padTop, padLeft, sums, static_cast<f16 *>(preWarpOut));
	} else {
		hipLaunchKernelGGL(warp_pack_kernel<bf16>, dim3(nb), dim3(256), 0, stream,
		    static_cast<const f16 *>(state), static_cast<const f16 *>(flow), frame, frameStride, static_cast<bf16 *>(out), outPitch, H,
		    W, PW, padTop, padLeft, sums, static_cast<f16 *>(preWarpOut));
	}
	hipCheckLaunch("warp_pack");
}

namespace {
TemporalGeom temporalGeom(int H, int W, const TemporalParams &tp) {
	TemporalGeom g{};
	g.HH = 4 * H;
	g.WW = 4 * W;
	g.window = tp.window;
	g.GH = g.GW = 1;
	if (tp.window > 0) {
		g.GH = (g.HH + tp.window - 1) / tp.window;
		g.GW = (g.WW + tp.window - 1) / tp.window;
		g.padY = (g.GH * tp.window - g.HH) / 2;
		g.padX = (g.GW * tp.window - g.WW) / 2;
	}
	g.strength = tp.strength;
	g.threshold = tp.threshold;
	g.gain = tp.gain;
	g.l2 = tp.l2;
	g.limit = tp.limit;
	g.luma = tp.luma;
	g.lrPixels = H * W;
	return g;
}
}  // namespace

std::size_t temporalAccWords(int H, int W, int window) {
	if (window <= 0) return 1;
	return static_cast<std::size_t>((4 * H + window - 1) / window) * ((4 * W + window - 1) / window);
}

void launchTemporalFilter(void *state, const void *preWarp, std::uint8_t *outU8,
    std::ptrdiff_t outStride, int H, int W, const unsigned *sums, unsigned long long *acc,
    const TemporalParams &tp, hipStream_t stream) {
	const TemporalGeom g = temporalGeom(H, W, tp);
	const size_t nPix = (size_t)g.HH * g.WW;
	hipLaunchKernelGGL(zero_words_kernel, dim3(blocksFor((size_t)g.GH * g.GW)), dim3(256), 0, stream, acc,
	    g.GH * g.GW);
	hipCheckLaunch("zero_words");
	hipLaunchKernelGGL(temporal_reduce_kernel, dim3(tp.window > 0 ? blocksFor(nPix) : 2048), dim3(256), 0,
	    stream, static_cast<const f16 *>(state), static_cast<const f16 *>(preWarp), g, sums, acc);
	hipCheckLaunch("temporal_reduce");
	hipLaunchKernelGGL(temporal_blend_kernel, dim3(blocksFor(nPix)), dim3(256), 0, stream,
	    static_cast<f16 *>(state), static_cast<const f16 *>(preWarp), outU8, outStride, g, sums, acc);
	hipCheckLaunch("temporal_blend");
}

void launchTail(DType dt, const void *y, const float *w2, const float *b2,
    const std::uint8_t *frame, std::ptrdiff_t frameStride, void *stateOut, std::uint8_t *outU8,
    std::ptrdiff_t outStride, int H, int W, const unsigned *sums, hipStream_t stream) {
	const unsigned nb = blocksFor((size_t)4 * H * W);
	if (dt == kF16) {
		hipLaunchKernelGGL(tail_kernel<f16>, dim3(nb), dim3(256), 0, stream,
		    static_cast<const f16 *>(y), w2, b2, frame, frameStride, static_cast<f16 *>(stateOut),
		    outU8, outStride, H, W, sums);
	} else {
		hipLaunchKernelGGL(tail_kernel<bf16>, dim3(nb), dim3(256), 0, stream,
		    static_cast<const bf16 *>(y), w2, b2, frame, frameStride, static_cast<f16 *>(stateOut),
		    outU8, outStride, H, W, sums);
	}
	hipCheckLaunch("tail");
}

void launchTailFused(DType dt, const TailFusedLaunch &q, hipStream_t stream) {
	TailFusedParams p{};
	p.x = q.x;
	p.xPitch = q.xPitch ? q.xPitch : q.W;
	p.w1 = q.w1;
	p.b1 = q.b1;
	p.w2 = q.w2;
	p.b2 = q.b2;
	p.frame = q.frame;
	p.frameStride = q.frameStride;
	p.state = q.state;
	p.outU8 = q.outU8;
	p.outStride = q.outStride;
	p.sums = q.sums;
	p.H = q.H;
	p.W = q.W;
	p.slope = q.slope;
	if (dt == kF16) launchTailFusedT<f16>(p, stream);
	else launchTailFusedT<bf16>(p, stream);
}

void launchFrameSums(const std::uint8_t *frame, std::ptrdiff_t frameStride, int H, int W,
    unsigned *sums, hipStream_t stream) {
	hipLaunchKernelGGL(frame_sums_kernel, dim3(1), dim3(1024), 0, stream, frame, frameStride, H, W,
	    sums);
	hipCheckLaunch("frame_sums");
}

void launchCopyRows(const std::uint8_t *src, std::ptrdiff_t srcStride, std::uint8_t *dst,
    std::ptrdiff_t dstStride, std::size_t rowBytes, std::size_t rows, hipStream_t stream) {
	const unsigned words = static_cast<unsigned>(rowBytes / 4);
	hipLaunchKernelGGL(copy_rows_kernel, dim3(blocksFor((size_t)words * rows)), dim3(256), 0,
	    stream, src, srcStride, dst, dstStride, words, static_cast<unsigned>(rows));
	hipCheckLaunch("copy_rows");
}

void launchSignalHost(unsigned *word, hipStream_t stream) {
	hipLaunchKernelGGL(signal_host_kernel, dim3(1), dim3(1), 0, stream, word);
	hipCheckLaunch("signal_host");
}

void launchToFloat(DType dt, const void *in, float *out, std::size_t n, hipStream_t stream) {
	if (dt == kF16) {
		hipLaunchKernelGGL(to_float_kernel<f16>, dim3(blocksFor(n)), dim3(256), 0, stream,
		    static_cast<const f16 *>(in), out, n);
	} else {
		hipLaunchKernelGGL(to_float_kernel<bf16>, dim3(blocksFor(n)), dim3(256), 0, stream,
		    static_cast<const bf16 *>(in), out, n);
	}
	hipCheckLaunch("to_float");
}


}  // namespace ju
