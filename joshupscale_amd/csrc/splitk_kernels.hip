// gfx950 (CDNA4, MI355X): the 3x3 convolutions of the COARSE levels of the flow auto-encoder
// (reference scripts/training/models.py:377-468; 68x120 and 34x60 at 480x270: 128-256 channels, a few
// thousand pixels), the input channels split over the eight waves of a workgroup.
// (moved out of flow_kernels.hip in round 4)
#include "flow_block_common.h"

namespace ju {

namespace {


// ---------------------------------------------------------------------------
// conv_splitk_kernel: one 3x3 convolution of the COARSE levels of the flow net
// (68x120 and 34x60 at 480x270: 128-256 channels, 2-8 k pixels), K split over the waves
// ---------------------------------------------------------------------------
// These layers are 1-5 GFLOP on a few thousand pixels: as conv_mfma_kernel launches they
// were latency chains (7-14 us each: per 64-channel chunk a global -> register -> LDS
// staging round for BOTH operands, 2-4 chunks in series, 40-140 workgroups).  Here one
// workgroup (8 waves) owns a tile of 32 columns x TH rows x 32*CB output channels and the
// eight waves split the INPUT channels: wave w keeps the A fragments of its CIN/8 channels
// (9 taps x KS k-steps x CB cout blocks) in registers for the whole tile, so every weight
// byte enters the CU once and all of it is in flight at once -- one load round trip instead
// of one per chunk.  The tile is a ring of four input rows per 64-channel plane (LDS-DMA,
// out-of-image pixels fetched from a zero page, so no fill pass): while a row pair's
// partial sums are reduced, the next pair's two rows land on the two rows it no longer
// needs.  Reduction: wave o owns piece o = (row o / 4, cout group o % 4) of the 32 x 32
// partial tile; every wave sends it the matching piece of its accumulators through LDS
// (7 KB per wave), the owner adds the eight pieces in wave order, applies bias and
// activation and puts its 4 values per lane into a staging tile, from which the
// workgroup stores whole 16-byte chunks (or their 2x2 maxima: POOL).
// fp32 accumulation per wave from zero, partial sums added in wave order, bias last: the
// summation order differs from conv_mfma_kernel's, nothing else (JU_FLOW_CONV=generic
// keeps that kernel; tests compare both).
struct SplitKParams {
	const void *in;     // NHWC [H][W][CIN]
	const void *wgt;    // packConvWeights(nb = 1): [COUT/32][CIN/64][9][4][2][32][8]
	const float *bias;  // [COUT]
	void *out;          // [H][W][COUT], or [H/2][W/2][COUT] with POOL
	const void *zeros;  // >= 16 zero bytes in device memory (source of out-of-image pixels)
	int H, W, cout;
	int inPitch, outPitch;
	int tilesX, TH;
	int act;
	float slope;
	int skip;  // timing ablation (JU_FB_SKIP, developer only): 1 weights, 2 staging, 4 K loops, 8 reduction, 16 stores
	int prio;  // wave priority scheme (kernel_common.h applyWavePriority)
	long inItem, outItem;  // frame look-ahead: blockIdx.z = the frame, its tensors at in + z * inItem / out + z * outItem bytes
};

template <int CIN, int CB>
struct SkGeom {
	static constexpr int NPL = CIN / 64;
	static constexpr int KS = CIN / 128;  // k-steps (16 channels) per wave
	static constexpr int XPLANE = 4 * kFbW * 128;
	static constexpr int OFF_P = NPL * XPLANE;
	static constexpr int PBYTES = CB * 8 * 7 * 1024;
	static constexpr int OFF_S = OFF_P + PBYTES;
	static constexpr int SBYTES = CB * 2 * 32 * 64;
	static constexpr int LDS = OFF_S + SBYTES;
	static_assert(CIN == 128 || CIN == 256, "input channels");
	static_assert(LDS <= 160 * 1024, "tile does not fit LDS");
};

template <typename T, int CIN, int CB, bool POOL>
__global__ __launch_bounds__(512, 1) void conv_splitk_kernel(SplitKParams p) {
	using G = SkGeom<CIN, CB>;
	constexpr int KS = G::KS;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = threadIdx.x;
	const int wave = tid >> 6;
	applyWavePriority(p.prio, wave, 8);
	const int lane = tid & 63;
	const int px = lane & 31;
	const int hh = lane >> 5;
	const int tx = blockIdx.x % p.tilesX, ty = blockIdx.x / p.tilesX;
	const int x0 = tx * 32, y0 = ty * p.TH;
	const int cog0 = blockIdx.y * CB;
	const int nPairs = (min(p.TH, p.H - y0) + 1) >> 1;
	const T *__restrict__ in = reinterpret_cast<const T *>(static_cast<const unsigned char *>(p.in) + blockIdx.z * p.inItem);
	const unsigned ldsBase = static_cast<unsigned>(reinterpret_cast<unsigned long long>(
	    (__attribute__((address_space(3))) unsigned char *)smem));
	// this wave's input channels: k-steps wave*KS .. +KS of the CIN/16
	const int plane = (wave * KS) >> 2;
	const int ksBase = (wave * KS) & 3;

	// ---- A fragments: CB cout blocks x 9 taps x KS k-steps, straight to registers ----
	Vec8<T> w[CB][9 * KS];
#pragma unroll
	for (int b = 0; b < CB; ++b) {
		const unsigned char *wsrc = static_cast<const unsigned char *>(p.wgt) +
		    ((size_t)((cog0 + b) * G::NPL + plane) * 9 * 4 + ksBase) * 1024 + lane * 16;
#pragma unroll
		for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
			for (int ks = 0; ks < KS; ++ks) {
				if (!(JU_SKIP(p) & 1)) w[b][tap * KS + ks] = *reinterpret_cast<const Vec8<T> *>(wsrc + (size_t)(tap * 4 + ks) * 1024);
				else w[b][tap * KS + ks] = Vec8<T>{};
			}
		}
	}
	// ---- tile rows [t0, t0 + nr) -> ring slots (t mod 4), all planes ----
	// tile row t = image row y0 - 1 + t, tile column k = image column x0 - 1 + k
	auto stageRows = [&](int t0, int nr) {
		if (JU_SKIP(p) & 2) return;
		const int nPix = nr * kFbW;
		const int nInstr = (nPix + 7) >> 3;  // 8 pixels of one plane per wave-instruction
		const int slot0 = t0 & 3;            // (t0 is even: the rows of a stage never wrap)
		for (int j = wave; j < G::NPL * nInstr; j += 8) {
			const int pl = j / nInstr, i = j - pl * nInstr;
			const int q = i * 8 + (lane >> 3);
			const int r = q / kFbW, k = q - r * kFbW;
			const int gy = y0 - 1 + t0 + r, gx = x0 - 1 + k;
			const unsigned c = static_cast<unsigned>(lane & 7) ^ fbSwz<128>(k);
			if (q < nPix) {
				const bool inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
				const void *src = inside ? static_cast<const void *>(in + ((size_t)gy * p.inPitch + gx) * CIN + pl * 64 + c * 8)
				                         : p.zeros;
				fbGlds16(src, smem + pl * G::XPLANE + slot0 * (kFbW * 128) + i * 1024);
			}
		}
	};
	stageRows(0, 4);

	unsigned colOff[3], colSwz[3];
#pragma unroll
	for (int dx = 0; dx < 3; ++dx) {
		colOff[dx] = (px + dx) * 128;
		colSwz[dx] = fbSwz<128>(px + dx);
	}
	const int hhx = ksBase * 2 + hh;  // chunk index of this wave's first k-step, this lane's half
	// owner role: piece (row orow, cout group og) of every cout block
	const int orow = wave >> 2, og = wave & 3;
	f32x4 biasv[CB];
#pragma unroll
	for (int b = 0; b < CB; ++b) {
		biasv[b] = *reinterpret_cast<const f32x4 *>(p.bias + (cog0 + b) * 32 + 8 * og + 4 * hh);
	}
	const float sAct = fbActS(p.act, p.slope);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();

	const unsigned xBase = ldsBase + plane * G::XPLANE;
	for (int pr = 0; pr < nPairs; ++pr) {
		f32x16 acc[CB][2];
#pragma unroll
		for (int b = 0; b < CB; ++b) {
#pragma unroll
			for (int r = 0; r < 2; ++r) {
#pragma unroll
				for (int i = 0; i < 16; ++i) acc[b][r][i] = 0.0f;
			}
		}
#pragma unroll
		for (int b = 0; b < (JU_SKIP(p) & 4 ? 0 : CB); ++b) {
			if (pr & 1) FbPair<T, KS, 128>::template run<2>(xBase, colOff, colSwz, hhx, w[b], acc[b]);
			else FbPair<T, KS, 128>::template run<0>(xBase, colOff, colSwz, hhx, w[b], acc[b]);
		}
		__syncthreads();  // B: rows 2pr, 2pr+1 are dead, the previous pair's pieces and staging tile too
		if (pr + 1 < nPairs) stageRows(2 * pr + 4, 2);
		// ---- pieces to their owners ----
#pragma unroll
		for (int b = 0; b < (JU_SKIP(p) & 8 ? 0 : CB); ++b) {
#pragma unroll
			for (int o = 0; o < 8; ++o) {
				if (o != wave) {
					const int slot = wave - (wave > o ? 1 : 0);
					const f32x4 v = {acc[b][o >> 2][4 * (o & 3)], acc[b][o >> 2][4 * (o & 3) + 1],
					    acc[b][o >> 2][4 * (o & 3) + 2], acc[b][o >> 2][4 * (o & 3) + 3]};
					*reinterpret_cast<f32x4 *>(smem + G::OFF_P + ((b * 8 + o) * 7 + slot) * 1024 + lane * 16) = v;
				}
			}
		}
		__syncthreads();  // C
		// ---- owner: sum in wave order, bias, activation, 16-bit, staging tile ----
#pragma unroll
		for (int b = 0; b < CB; ++b) {
			f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
			for (int src = 0; src < (JU_SKIP(p) & 8 ? 0 : 8); ++src) {
				f32x4 v;
				if (src == wave) {
					// (this wave's own piece: select by the uniform owner index)
					f32x4 own = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
					for (int o = 0; o < 8; ++o) {
						if (o == wave) {
							own = f32x4{acc[b][o >> 2][4 * (o & 3)], acc[b][o >> 2][4 * (o & 3) + 1],
							    acc[b][o >> 2][4 * (o & 3) + 2], acc[b][o >> 2][4 * (o & 3) + 3]};
						}
					}
					v = own;
				} else {
					const int slot = src - (src > wave ? 1 : 0);
					v = *reinterpret_cast<const f32x4 *>(smem + G::OFF_P + ((b * 8 + wave) * 7 + slot) * 1024 + lane * 16);
				}
				sum += v;
			}
			const Vec4<T> o16 = pack4<T>(fbAct(sum[0] + biasv[b][0], sAct), fbAct(sum[1] + biasv[b][1], sAct),
			    fbAct(sum[2] + biasv[b][2], sAct), fbAct(sum[3] + biasv[b][3], sAct));
			// staging tile [b][row][px][32 couts], the 16-byte chunk index swizzled by the column
			*reinterpret_cast<Vec4<T> *>(smem + G::OFF_S + ((b * 2 + orow) * 32 + px) * 64 +
			    ((static_cast<unsigned>(og) ^ ((px >> 2) & 3u)) << 4) + hh * 8) = o16;
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the next pair's rows)
		__syncthreads();  // D
		// ---- staging tile -> global, 16 bytes per thread ----
		T *out = reinterpret_cast<T *>(static_cast<unsigned char *>(p.out) + blockIdx.z * p.outItem);
		if constexpr (!POOL) {
			if (tid < CB * 256) {
				const int q = tid & 3, cx = (tid >> 2) & 31, r = (tid >> 7) & 1, b = tid >> 8;
				const int gy = y0 + 2 * pr + r, gx = x0 + cx;
				if (gy < p.H && gx < p.W && !(JU_SKIP(p) & 16)) {
					const uint4 v = *reinterpret_cast<const uint4 *>(smem + G::OFF_S + ((b * 2 + r) * 32 + cx) * 64 +
					    ((static_cast<unsigned>(q) ^ ((cx >> 2) & 3u)) << 4));
					*reinterpret_cast<uint4 *>(out + ((size_t)gy * p.outPitch + gx) * p.cout + (cog0 + b) * 32 + q * 8) = v;
				}
			}
		} else {
			if (tid < CB * 64) {
				const int q = tid & 3, kx = (tid >> 2) & 15, b = tid >> 6;
				const int gy = (y0 >> 1) + pr, gx = (x0 >> 1) + kx;
				if (gy < (p.H >> 1) && gx < (p.W >> 1)) {
					Vec8<T> m;
#pragma unroll
					for (int r = 0; r < 2; ++r) {
#pragma unroll
						for (int d = 0; d < 2; ++d) {
							const int cx = 2 * kx + d;
							const Vec8<T> v = *reinterpret_cast<const Vec8<T> *>(smem + G::OFF_S + ((b * 2 + r) * 32 + cx) * 64 +
							    ((static_cast<unsigned>(q) ^ ((cx >> 2) & 3u)) << 4));
#pragma unroll
							for (int j = 0; j < 8; ++j) {
								m[j] = (r == 0 && d == 0) ? v[j] : (static_cast<float>(v[j]) > static_cast<float>(m[j]) ? v[j] : m[j]);
							}
						}
					}
					*reinterpret_cast<Vec8<T> *>(out + ((size_t)gy * p.outPitch + gx) * p.cout + (cog0 + b) * 32 + q * 8) = m;
				}
			}
		}
	}
}

template <typename T, int CIN, int CB, bool POOL>
void launchSplitKInst(const SplitKParams &p, int tilesY, int items, hipStream_t stream) {
	using G = SkGeom<CIN, CB>;
	auto kern = conv_splitk_kernel<T, CIN, CB, POOL>;
	static std::atomic<std::uint64_t> ldsDone{0};
	ensureDynamicLds(reinterpret_cast<const void *>(kern), G::LDS, &ldsDone, "split-K conv");
	if (launchesAreDry()) return;
	hipLaunchKernelGGL(kern, dim3(p.tilesX * tilesY, p.cout / (32 * CB), items), dim3(512), G::LDS, stream, p);
	hipCheckLaunch("conv_splitk");
}

}  // namespace


bool convSplitKSupported(const ConvParams &p) {
	// (a pooled 256-channel layer does not occur in the flow net: not instantiated, not claimed)
	return p.taps == 9 && (p.cin == 128 || p.cin == 256) && p.cout % 32 == 0 && !p.res && !p.outHead && !p.upsample &&
	       p.nb == 1 && (!p.pool || (p.cin == 128 && p.H % 2 == 0 && p.W % 2 == 0)) && p.H * p.W <= 32768;
}

void launchConvSplitK(DType dt, const ConvParams &q, const void *zeros, hipStream_t stream) {
	if (!convSplitKSupported(q) || !zeros) throw std::invalid_argument("split-K conv: unsupported layer");
	const int cus = currentDeviceCUs();
	SplitKParams p{};
	p.in = q.in;
	p.wgt = q.wgt;
	p.bias = q.bias;
	p.out = q.out;
	p.zeros = zeros;
	p.H = q.H;
	p.W = q.W;
	p.cout = q.cout;
	p.inPitch = q.inPitch ? q.inPitch : q.W;
	p.outPitch = q.outPitch ? q.outPitch : (q.pool ? q.W / 2 : q.W);
	p.act = q.relu;
	p.slope = q.slope;
	p.skip = ablationSkipBits();
	p.prio = wavePriorityMode(0);
	p.tilesX = (q.W + 31) / 32;
	const int items = q.items > 1 ? q.items : 1;  // (frame look-ahead: the layer of `items` frames in one launch)
	p.inItem = items > 1 ? q.inItemBytes : 0;
	p.outItem = items > 1 ? q.outItemBytes : 0;
	// Tile height and cout blocks per workgroup: every workgroup pulls its cout blocks'
	// whole weights (147 KB per block at 256 channels), so the fewest workgroups that still
	// fill most of the chip in ONE round; two cout blocks per workgroup (128 channels only:
	// LDS) when even the tallest tile leaves more workgroups than CUs.
	const int nCog = q.cout / 32;
	int cb = 1, th = 2;
	if (items > 1) {
		// A look-ahead launch has several rounds' worth of tiles: the height (and cout blocks per workgroup) whose
		// rounds x (rows + a tile's fixed part: weights, ramp -- about six rows' worth) is smallest.  34 x 60, 256
		// couts, 8 frames: 18-row tiles are exactly one round (2 x 2 x 8 x 8 = 256 workgroups); the one-frame rule
		// below would stop at 16 rows and pay 1.5 rounds (53 -> 33 us per pass, profiles/r05_flow_layers_pass.txt).
		long best = -1;
		for (int c = 1; c <= ((q.cin == 128 && nCog % 2 == 0) ? 2 : 1); ++c) {
			for (int t = 2; t <= 34; t += 2) {
				const long wgs = (long)p.tilesX * ((q.H + t - 1) / t) * (nCog / c) * items;
				const long cost = (wgs + cus - 1) / cus * c * (std::min(t, (q.H + 1) / 2 * 2) + 6);
				if (best < 0 || cost < best) {
					best = cost;
					cb = c;
					th = t;
				}
			}
		}
	} else {
		for (;;) {
			bool found = false;
			for (th = 2; th <= 16; th += 2) {
				if ((long)p.tilesX * ((q.H + th - 1) / th) * (nCog / cb) <= cus) {
					found = true;
					break;
				}
			}
			if (found || cb == 2 || q.cin != 128 || nCog % 2) break;
			cb = 2;
		}
		if (th > 16) th = 16;
	}
	p.TH = th;
	const int tilesY = (q.H + th - 1) / th;
	const bool f16t = dt == kF16;
#define JU_SK_CASE(CIN_, CB_, POOL_)                                                       \
	if (q.cin == CIN_ && cb == CB_ && (q.pool != 0) == POOL_) {                              \
		if (f16t) launchSplitKInst<f16, CIN_, CB_, POOL_>(p, tilesY, items, stream);           \
		else launchSplitKInst<bf16, CIN_, CB_, POOL_>(p, tilesY, items, stream);               \
		return;                                                                              \
	}
	JU_SK_CASE(128, 1, false)
	JU_SK_CASE(128, 1, true)
	JU_SK_CASE(128, 2, false)
	JU_SK_CASE(128, 2, true)
	JU_SK_CASE(256, 1, false)
#undef JU_SK_CASE
	throw std::invalid_argument("split-K conv: unsupported shape");
}


}  // namespace ju
