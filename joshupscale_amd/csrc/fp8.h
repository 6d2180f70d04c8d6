// 8-bit (OCP e4m3fn) form of the generator's residual tower: host-side quantiser
// and operand packing for conv_tower_fp8_kernel (fp8_kernels.hip).
//
// The reference's 8-bit deployment is TensorRT INT8 with symmetric per-tensor
// activation scales and per-channel weight scales
// (scripts/inference/tensorrt/quantize_int8.py:140-209, generate_calibration.py:93-234);
// BASELINE.json config 5 asks for the fp8-MFMA counterpart on gfx950.  Scheme:
//   * weights of the 48 block convolutions: per OUTPUT channel c a power-of-two scale
//     2^ew[c] with max|w[c]| * 2^ew[c] in (224, 448], stored as e4m3 (round to
//     nearest even);
//   * the input of every block convolution (a post-ReLU tensor): one power-of-two
//     scale 2^ea per tensor, stored as e4m3(min(x * 2^ea, 448));
//   * products accumulate in fp32 on the matrix cores; both scales are undone by the
//     block-scaled MFMA's E8M0 scale operands (exact: powers of two);
//   * bias, ReLU and the skip connection stay fp32 / 16-bit: the residual stream is
//     never quantised to 8 bits.
#pragma once

#include <cmath>
#include <cstdint>
#include <vector>

#include "model.h"

namespace ju {

// Round to nearest even, saturating at +-448 (the hardware conversion
// v_cvt_pk_fp8_f32 agrees on every in-range input: tools/probes/fp8_mfma_probe.hip).
inline std::uint8_t e4m3FromFloat(float x) {
	if (std::isnan(x)) return 0x7f;
	const std::uint8_t s = std::signbit(x) ? 0x80 : 0;
	const float a = std::fabs(x);
	if (a >= 448.0f) return s | 0x7e;
	int e = 0;
	std::frexp(a, &e);
	int E = e - 1;  // a = 1.xxx * 2^E
	if (E < -6) E = -6;  // subnormals: step 2^-9
	const float step = std::ldexp(1.0f, E - 3);
	const float v = std::nearbyint(a / step) * step;  // ties to even (default rounding mode)
	if (v >= 448.0f) return s | 0x7e;
	if (v < std::ldexp(1.0f, -6)) {
		return s | static_cast<std::uint8_t>(std::lround(v * 512.0f));
	}
	std::frexp(v, &e);
	const int E2 = e - 1;
	const int mant = static_cast<int>(std::lround((std::ldexp(v, -E2) - 1.0f) * 8.0f));
	return s | static_cast<std::uint8_t>(((E2 + 7) << 3) | mant);
}

// Exponent of the power-of-two scale that maps [0, amax] into e4m3 with one bit of
// headroom (amax * 2^e in (112, 224]); clamped so that the E8M0 code stays valid.
inline int fp8ActivationExponent(float amax) {
	if (!(amax > 0.0f) || !std::isfinite(amax)) return 0;
	int e = static_cast<int>(std::floor(std::log2(224.0f / amax)));
	return e < -16 ? -16 : (e > 16 ? 16 : e);
}

// Range assumed for a block-convolution input when the model carries no
// "generator/fp8_amax" tensor: e4m3 is a floating-point format, so one fixed scale
// (2^5: full 3-bit mantissa for 0.0005 <= x <= 14) serves any sane post-BN tensor.
constexpr float kFp8DefaultAmax = 7.0f;

struct Fp8ConvWeights {
	std::vector<std::uint8_t> w;   // [tap 9][cout block 2][lane 64][32 bytes]: A fragments
	std::vector<std::int32_t> scaleA;  // [64] E8M0 code of 2^-ew[c]
};

// f: folded 3x3 64->64 convolution, w = [tap][cin][cout].  Lane (r = lane & 31,
// h = lane >> 5) of fragment (tap, nb) holds input channels 32h .. 32h+31 of output
// channel 32nb + r: the byte order a lane reads from a 64-byte e4m3 pixel record.
inline Fp8ConvWeights packFp8TowerWeights(const FoldedConv &f) {
	Fp8ConvWeights out;
	out.w.assign(9 * 2 * 64 * 32, 0);
	out.scaleA.assign(64, 127);
	for (int c = 0; c < 64; ++c) {
		float amax = 0.0f;
		for (int t = 0; t < 9; ++t) {
			for (int k = 0; k < 64; ++k) amax = std::fmax(amax, std::fabs(f.w[(t * 64 + k) * 64 + c]));
		}
		int ew = 0;
		if (amax > 0.0f && std::isfinite(amax)) {
			ew = static_cast<int>(std::floor(std::log2(448.0f / amax)));
			ew = ew < -32 ? -32 : (ew > 32 ? 32 : ew);
		}
		out.scaleA[c] = 127 - ew;
		const int nb = c >> 5, r = c & 31;
		for (int t = 0; t < 9; ++t) {
			for (int k = 0; k < 64; ++k) {
				const int lane = (k >> 5) * 32 + r;
				out.w[((t * 2 + nb) * 64 + lane) * 32 + (k & 31)] =
				    e4m3FromFloat(std::ldexp(f.w[(t * 64 + k) * 64 + c], ew));
			}
		}
	}
	return out;
}

}  // namespace ju
