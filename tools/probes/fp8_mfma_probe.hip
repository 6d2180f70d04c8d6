// GPU probe (gfx950): operand layout and scale semantics of
// v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands, and rounding / saturation of
// v_cvt_pk_fp8_f32, checked with exact data against a host restatement.
// build: hipcc --offload-arch=gfx950 -O2 -o build/fp8_probe tools/probes/fp8_mfma_probe.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// OCP e4m3fn, round to nearest even, saturating to +-448
static uint8_t e4m3_from_float(float x) {
	if (std::isnan(x)) return 0x7f;
	uint8_t s = std::signbit(x) ? 0x80 : 0;
	float a = std::fabs(x);
	if (a >= 448.0f) return s | 0x7e;
	if (a < std::ldexp(1.0f, -10)) return s;  // below half of the smallest subnormal
	int e;
	std::frexp(a, &e);  // a = m * 2^e, m in [0.5,1)
	int E = e - 1;      // a = 1.xxx * 2^E
	if (E < -6) E = -6;  // subnormal: step 2^-9
	const float step = std::ldexp(1.0f, E - 3);
	float q = std::nearbyint(a / step);  // RNE in the default rounding mode
	float v = q * step;
	if (v >= 448.0f) return s | 0x7e;
	std::frexp(v, &e);
	int E2 = e - 1;
	if (v < std::ldexp(1.0f, -6)) {  // subnormal
		return s | static_cast<uint8_t>(std::lround(v / std::ldexp(1.0f, -9)));
	}
	int mant = static_cast<int>(std::lround((v / std::ldexp(1.0f, E2) - 1.0f) * 8.0f));
	return s | static_cast<uint8_t>(((E2 + 7) << 3) | mant);
}
static float e4m3_to_float(uint8_t b) {
	const int e = (b >> 3) & 15, m = b & 7;
	float v = e == 0 ? std::ldexp(static_cast<float>(m), -9) : std::ldexp(1.0f + m / 8.0f, e - 7);
	if (e == 15 && m == 7) v = NAN;
	return (b & 0x80) ? -v : v;
}

__global__ void mfma_kernel(const uint8_t *A, const uint8_t *B, const int *sa, const int *sb, float *D) {
	const int lane = threadIdx.x;
	v8i a, b;
	for (int i = 0; i < 8; ++i) {
		a[i] = reinterpret_cast<const int *>(A)[lane * 8 + i];
		b[i] = reinterpret_cast<const int *>(B)[lane * 8 + i];
	}
	f32x16 c = {};
	c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa[lane], 0, sb[lane]);
	for (int i = 0; i < 16; ++i) D[lane * 16 + i] = c[i];
}

__global__ void cvt_kernel(const float *x, uint8_t *y, int n) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i * 2 + 1 < n) {
		int packed = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0, false);
		y[2 * i] = packed & 0xff;
		y[2 * i + 1] = (packed >> 8) & 0xff;
	}
}

int main() {
	std::mt19937 rng(7);
	// ---- MFMA: hypothesis  lane (r = l&31, h = l>>5), byte j  <->  k = 32h + j  for A[r][k] and B[k][r];
	//      scale of lane l applies to (row/col l&31, k block l>>5):  value = 2^(s-127)
	std::vector<float> Am(32 * 64), Bm(64 * 32);
	std::vector<uint8_t> Af(64 * 32), Bf(64 * 32);
	std::vector<int> sa(64), sb(64);
	for (auto &v : Am) v = static_cast<float>(static_cast<int>(rng() % 9) - 4);
	for (auto &v : Bm) v = static_cast<float>(static_cast<int>(rng() % 9) - 4) * 0.5f;
	for (int l = 0; l < 64; ++l) {
		// per row / per column scales, equal for both k blocks (what the convolution uses:
		// with lane-half-dependent scales the k-block <-> lane mapping is NOT "bytes of lane
		// half h", measured 787/1024 mismatches)
		sa[l] = 127 + ((l & 31) * 7 + 3) % 5 - 2;
		sb[l] = 127 + ((l & 31) * 5 + 1) % 5 - 2;
		for (int j = 0; j < 32; ++j) {
			Af[l * 32 + j] = e4m3_from_float(Am[(l & 31) * 64 + 32 * (l >> 5) + j]);
			Bf[l * 32 + j] = e4m3_from_float(Bm[(32 * (l >> 5) + j) * 32 + (l & 31)]);
		}
	}
	uint8_t *dA, *dB;
	int *dsa, *dsb;
	float *dD;
	hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dD, 4096);
	hipMemcpy(dA, Af.data(), 2048, hipMemcpyHostToDevice);
	hipMemcpy(dB, Bf.data(), 2048, hipMemcpyHostToDevice);
	std::vector<float> D(1024);
	for (int mode = 0; mode < 4; ++mode) {
		std::vector<int> ua(64, 127), ub(64, 127);
		if (mode & 1) ua = sa;
		if (mode & 2) ub = sb;
		hipMemcpy(dsa, ua.data(), 256, hipMemcpyHostToDevice);
		hipMemcpy(dsb, ub.data(), 256, hipMemcpyHostToDevice);
		hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
		hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
		int bad = 0;
		for (int l = 0; l < 64; ++l) {
			for (int reg = 0; reg < 16; ++reg) {
				const int col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
				double ref = 0;
				for (int k = 0; k < 64; ++k) {
					const int kb = k >> 5;
					ref += static_cast<double>(Am[row * 64 + k]) * Bm[k * 32 + col] *
					       std::ldexp(1.0, ua[kb * 32 + row] - 127) * std::ldexp(1.0, ub[kb * 32 + col] - 127);
				}
				if (std::fabs(ref - D[l * 16 + reg]) > 1e-4) {
					if (bad < 3) printf("mismatch lane %d reg %d: got %g want %g\n", l, reg, D[l * 16 + reg], ref);
					++bad;
				}
			}
		}
		printf("mfma_scale 32x32x64 e4m3, scales %s%s: %d mismatches of 1024\n", (mode & 1) ? "A " : "", (mode & 2) ? "B" : "", bad);
	}
	// ---- pairing of A and B fragment bytes (unit scales): A one-hot at (h, j) of row 0,
	//      B column 0 holds 64 distinct values, D[0][0] names the B byte it met
	{
		std::vector<int> one(64, 127);
		hipMemcpy(dsa, one.data(), 256, hipMemcpyHostToDevice);
		hipMemcpy(dsb, one.data(), 256, hipMemcpyHostToDevice);
		std::vector<uint8_t> Bz(2048, 0);
		for (int h = 0; h < 2; ++h) for (int j = 0; j < 32; ++j) Bz[(h * 32 + 0) * 32 + j] = 0x08 + h * 32 + j;
		hipMemcpy(dB, Bz.data(), 2048, hipMemcpyHostToDevice);
		int same = 0;
		for (int h = 0; h < 2; ++h) {
			for (int j = 0; j < 32; ++j) {
				std::vector<uint8_t> Az(2048, 0);
				Az[(h * 32 + 0) * 32 + j] = 0x38;  // 1.0 in lane (row 0, half h), byte j
				hipMemcpy(dA, Az.data(), 2048, hipMemcpyHostToDevice);
				hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
				hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
				const uint8_t code = e4m3_from_float(D[0]);
				const int hb = (code - 0x08) >> 5, jb = (code - 0x08) & 31;
				if (hb == h && jb == j) ++same; else printf("A(h=%d,j=%d) pairs with B(h=%d,j=%d)\n", h, j, hb, jb);
			}
		}
		printf("A/B byte pairing identical for %d of 64 positions\n", same);
		// scale semantics: A = B = ones on k-block pattern; lane-varying scale_a only
		std::vector<uint8_t> ones(2048, 0x38);
		hipMemcpy(dA, ones.data(), 2048, hipMemcpyHostToDevice);
		hipMemcpy(dB, ones.data(), 2048, hipMemcpyHostToDevice);
		std::vector<int> sv(64);
		for (int l = 0; l < 64; ++l) sv[l] = 127 + (l & 3) + 4 * (l >> 5);  // row-dependent, half-dependent
		hipMemcpy(dsa, sv.data(), 256, hipMemcpyHostToDevice);
		hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
		hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
		printf("scale_a = 127 + (lane&3) + 4*(lane>>5), unit B scale: D[row][col 0] for rows 0..7:");
		for (int row = 0; row < 8; ++row) {
			const int l = 0 + 32 * ((row >> 2) & 1), reg = (row & 3) + 4 * (row >> 3);
			printf(" %g", D[l * 16 + reg]);
		}
		printf("\n  (32*2^(r&3) + 32*2^(4+(r&3)) = 544, 1088, 2176, 4352 if lane l scales row l&31, k block l>>5)\n");
		printf("  D[0][col 0..3]: %g %g %g %g\n", D[0 * 16], D[1 * 16], D[2 * 16], D[3 * 16]);
	}

	// ---- cvt_pk_fp8_f32 against the host quantizer
	const int n = 1 << 16;
	std::vector<float> x(n);
	std::uniform_real_distribution<float> u(-1.f, 1.f);
	for (int i = 0; i < n; ++i) {
		const int kind = i & 7;
		const float m = u(rng);
		x[i] = kind == 0 ? m * 600.f : kind == 1 ? m * 0.02f : kind == 2 ? m * 0.002f : kind == 3 ? m * 30.f : m * 2.f;
	}
	// exact ties and edges
	const float edges[] = {448.f, 464.f, 463.9f, 480.f, 1000.f, 1e30f, INFINITY, -INFINITY, 0.f, -0.f,
	    0.001953125f, 0.0009765625f, 0.00097f, 0.0029296875f, 1.0625f, 1.1875f, 17.f, 19.f, 432.f, 447.9f};
	for (size_t i = 0; i < sizeof(edges) / 4; ++i) x[i] = edges[i];
	float *dx;
	uint8_t *dy;
	hipMalloc(&dx, n * 4); hipMalloc(&dy, n);
	hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(cvt_kernel, dim3(n / 2 / 256), dim3(256), 0, 0, dx, dy, n);
	std::vector<uint8_t> y(n);
	hipMemcpy(y.data(), dy, n, hipMemcpyDeviceToHost);
	int cbad = 0, sat = 0;
	for (int i = 0; i < n; ++i) {
		const uint8_t want = e4m3_from_float(x[i]);
		if (want != y[i]) {
			if (std::fabs(x[i]) >= 448.f) { ++sat; if (sat <= 6) printf("overflow %g -> 0x%02x (%g), saturating host 0x%02x\n", x[i], y[i], e4m3_to_float(y[i]), want); continue; }
			if (cbad < 8) printf("cvt mismatch %.9g: got 0x%02x (%g) want 0x%02x (%g)\n", x[i], y[i], e4m3_to_float(y[i]), want, e4m3_to_float(want));
			++cbad;
		}
	}
	printf("cvt_pk_fp8_f32: %d in-range mismatches of %d, %d overflow inputs not saturated\n", cbad, n, sat);
	return 0;
}
