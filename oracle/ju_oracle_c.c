/*
 * ju_oracle_c.c -- plain C (float32, OpenMP) restatement of the reference's
 * per-frame recurrent super-resolution step.
 *
 * TEST INFRASTRUCTURE / CPU BASELINE ONLY.  Nothing in the product path
 * (joshupscale_amd, libJoshUpscale.so) links, loads or calls this file; it is
 * used by tests/ as a second checker next to oracle/ju_oracle.py and by
 * bench.py's `cpu_baseline` leg as the timed stand-in for the reference's CPU
 * executors (onnxruntime CPU / Keras on tensorflow-cpu), which cannot run here.
 *
 * PARITY UNPINNED: the reference ships no golden vectors and cannot be built or
 * imported in this environment (see oracle/ju_oracle.py); this file is pinned
 * only against that numpy restatement (tests/test_oracle_cross.py, tests/test_golden.py).
 *
 * It follows the reference graph op for op and in the reference's order --
 * Conv2D, then BatchNormalization (NOT folded), then the activation -- in
 * float32, which is the precision the reference's Keras graph computes in:
 *   get_inference_model   scripts/training/models.py:680-829
 *   get_flow_autoencoder  scripts/training/models.py:334-481
 *   get_flow_resnet       scripts/training/models.py:257-331
 *   get_generator_resnet  scripts/training/models.py:484-595
 *   res_block             scripts/training/models.py:193-254
 *   dense_image_warp      scripts/training/tfa/dense_image_warp.py:87-245
 *   UpscaleLayer etc.     scripts/training/keras_layers.py:12-230
 * normalize_brightness (models.py:772-779, 802-803, 809-810) is followed; the optional temporal output filter is not
 * (a model that has it is refused).
 * It reads the same .jupw container as the engine (own parser, shares no code
 * with joshupscale_amd/csrc/model.cpp).
 *
 * Build: gcc -O3 -march=x86-64-v3 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define MAX_TENSORS 1024

typedef struct {
	char name[93];
	int ndim;
	int dims[4];
	const float *data;
	size_t count;
} juo_tensor;

typedef struct {
	unsigned char *blob;
	int H, W, PH, PW, n_in, arch, pad, gen_filters, gen_blocks, res_filters, res_blocks;
	int n_ff, ff[8];
	float eps;
	/* activation of the flow net / generator (models.py:24-27): negative slope, 0 = ReLU */
	float flow_slope, gen_slope;
	int n_tensors;
	juo_tensor t[MAX_TENSORS];
	/* recurrent state (zero-initialised: reference cuda.h:69-72) */
	int normalize;        /* normalize_brightness (models.py:772-779, 802-803, 809-810) */
	float *pre_gen;       /* [4H][4W][3] */
	float *last[8];       /* n_in-1 x [PH][PW][3] */
	float *output_raw;    /* last output, [4H][4W][3] */
} juo;

static uint32_t rd32(const unsigned char *p) {
	uint32_t v;
	memcpy(&v, p, 4);
	return v;
}
static uint64_t rd64(const unsigned char *p) {
	uint64_t v;
	memcpy(&v, p, 8);
	return v;
}

static const juo_tensor *get(const juo *m, const char *name) {
	for (int i = 0; i < m->n_tensors; ++i) {
		if (strcmp(m->t[i].name, name) == 0) return &m->t[i];
	}
	fprintf(stderr, "ju_oracle_c: missing tensor %s\n", name);
	abort();
}

static const float *getf(const juo *m, const char *prefix, const char *suffix) {
	char buf[200];
	snprintf(buf, sizeof(buf), "%s%s", prefix, suffix);
	return get(m, buf)->data;
}

int juo_num_threads(void) {
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}

/* The caller knows how many CPUs the process may really use (a container's CPU quota is invisible to
 * OpenMP, which counts the host's logical CPUs: 128 threads under a 16-CPU quota ran four times SLOWER
 * than 16 threads, profiles/r04_cpu_scaling.txt). */
void juo_set_num_threads(int n) {
#ifdef _OPENMP
	if (n >= 1) omp_set_num_threads(n);
#else
	(void)n;
#endif
}

/* ---- primitives ------------------------------------------------------- */

/* Conv2D(k x k, stride 1, "same"), NHWC, kernel [k][k][cin][cout], optional bias
 * (models.py:218-225).  cout <= 256. */
/* One output pixel, every output channel: the plain form (border columns, and widths / channel counts
 * the blocked form below does not take). */
static void conv_pixel(const float *x, int H, int W, int cin, const float *k, int ks, int cout,
    const float *bias, int h, int w, float *out) {
	const int p = (ks - 1) / 2;
	float acc[1024];
	for (int o = 0; o < cout; ++o) acc[o] = bias ? bias[o] : 0.0f;
	for (int a = 0; a < ks; ++a) {
		const int yy = h + a - p;
		if (yy < 0 || yy >= H) continue;
		for (int b = 0; b < ks; ++b) {
			const int xx = w + b - p;
			if (xx < 0 || xx >= W) continue;
			const float *xp = x + ((size_t)yy * W + xx) * cin;
			const float *kp = k + (size_t)(a * ks + b) * cin * cout;
			for (int c = 0; c < cin; ++c) {
				const float xv = xp[c];
				const float *kr = kp + (size_t)c * cout;
				for (int o = 0; o < cout; ++o) acc[o] += xv * kr[o];
			}
		}
	}
	memcpy(out, acc, sizeof(float) * cout);
}

/* layers.Conv2D(strides=1, padding="same"), odd square kernels (models.py:218-225, 300-306, 378-385,
 * 469-475, 531-537): y[h,w,o] = bias[o] + sum_{a,b,c} x[h+a-p, w+b-p, c] * K[a,b,c,o], taps outside
 * the image skipped.  Interior columns are computed A FEW PIXELS x 16 or 32 OUTPUT CHANNELS at a time, so
 * that a row of the kernel is loaded once for the block instead of once per pixel (the plain form
 * streams the whole kernel, 147 KB for 64 -> 64, through the cache for every pixel).  Every output
 * element still accumulates bias, then (a, b, c) in that order, with one multiply and one add per
 * term (the build sets -ffp-contract=off; the vector types below are element-wise): the results are
 * bit-identical to the plain form whatever the block shape and the vector width, which the tests
 * check.  The library is built for x86-64-v3 (it travels to another host); where the CPU has AVX-512
 * the 16-wide blocks are picked at run time. */
typedef float juo_v8 __attribute__((vector_size(32)));
typedef float juo_v16 __attribute__((vector_size(64)));
typedef float juo_v8_mem __attribute__((vector_size(32), aligned(4), may_alias));  /* as it lies in a tensor */
typedef float juo_v16_mem __attribute__((vector_size(64), aligned(4), may_alias));

/* PB pixels (columns w .. w+PB-1 of row h, all of whose taps' columns are inside the image) x TWO vectors of
 * output channels from `ob`.  The accumulators are named variables, not an array: gcc keeps an array of vectors
 * in memory and stores every update. */
#define JUO_REP1(M) M(0)
#define JUO_REP6(M) M(0) M(1) M(2) M(3) M(4) M(5)
#define JUO_REP8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define JUO_DECL(q) VT a##q##_0 = b0, a##q##_1 = b1;
#define JUO_STEP(q)                                  \
	{                                                \
		const float xv = xp[(size_t)(q) * cin + c];  \
		a##q##_0 += xv * k0;                         \
		a##q##_1 += xv * k1;                         \
	}
#define JUO_STORE(q)                                                \
	{                                                               \
		float *yp = y + ((size_t)h * W + w + (q)) * cout + ob;      \
		*(VTM *)yp = a##q##_0;                                      \
		*(VTM *)(yp + VL) = a##q##_1;                               \
	}
#define JUO_DEFINE_BLOCK(NAME, VEC, VECM, VLEN, REP, ATTR)                                              \
	ATTR static void NAME(const float *x, int H, int W, int cin, const float *k, int ks, int cout,      \
	    const float *bias, float *y, int h, int w, int ob) {                                             \
		typedef VEC VT;                                                                                 \
		typedef VECM VTM;                                                                               \
		enum { VL = VLEN };                                                                             \
		const int p = (ks - 1) / 2;                                                                     \
		VT b0 = {0}, b1 = {0};                                                                          \
		if (bias) {                                                                                     \
			b0 = *(const VTM *)(bias + ob);                                                             \
			b1 = *(const VTM *)(bias + ob + VL);                                                        \
		}                                                                                               \
		REP(JUO_DECL)                                                                                   \
		for (int a = 0; a < ks; ++a) {                                                                  \
			const int yy = h + a - p;                                                                   \
			if (yy < 0 || yy >= H) continue;                                                            \
			for (int b = 0; b < ks; ++b) {                                                              \
				const float *xp = x + ((size_t)yy * W + (w + b - p)) * cin;                             \
				const float *kp = k + (size_t)(a * ks + b) * cin * cout + ob;                           \
				for (int c = 0; c < cin; ++c) {                                                         \
					const VT k0 = *(const VTM *)(kp + (size_t)c * cout);                                \
					const VT k1 = *(const VTM *)(kp + (size_t)c * cout + VL);                           \
					REP(JUO_STEP)                                                                       \
				}                                                                                       \
			}                                                                                           \
		}                                                                                               \
		REP(JUO_STORE)                                                                                  \
	}

typedef void (*juo_block_fn)(const float *, int, int, int, const float *, int, int, const float *, float *, int, int, int);
JUO_DEFINE_BLOCK(conv_block_v8, juo_v8, juo_v8_mem, 8, JUO_REP6, )    /* 6 pixels x 16 channels: 12 + 2 + 1 of 16 registers */
JUO_DEFINE_BLOCK(conv_block1_v8, juo_v8, juo_v8_mem, 8, JUO_REP1, )
#if defined(__x86_64__)
JUO_DEFINE_BLOCK(conv_block_v16, juo_v16, juo_v16_mem, 16, JUO_REP8, __attribute__((target("avx512f"))))  /* 8 x 32 */
JUO_DEFINE_BLOCK(conv_block1_v16, juo_v16, juo_v16_mem, 16, JUO_REP1, __attribute__((target("avx512f"))))
#endif

static juo_block_fn g_block = conv_block_v8, g_block1 = conv_block1_v8;
static int g_block_px = 6, g_block_ch = 16, g_vector_bits = 256;

/* 512, 256 or 0: which form runs.  JUO_VECTOR_BITS=256 forces the narrow blocks, JUO_VECTOR_BITS=0 the plain
 * form for every pixel: tests/test_oracle_cross.py compares the bytes of the three. */
int juo_vector_bits(void) {
	static int chosen;
	if (!chosen) {
		chosen = 1;
		const char *e = getenv("JUO_VECTOR_BITS");
		if (e && atoi(e) == 0) {
			g_block_ch = 1 << 30;  /* no channel count is a multiple: conv_row takes the plain form */
			g_vector_bits = 0;
			return 0;
		}
#if defined(__x86_64__)
		if (__builtin_cpu_supports("avx512f") && !(e && atoi(e) == 256)) {
			g_block = conv_block_v16;
			g_block1 = conv_block1_v16;
			g_block_px = 8;
			g_block_ch = 32;
			g_vector_bits = 512;
		}
#endif
	}
	return g_vector_bits;
}

static void conv_row(const float *x, int H, int W, int cin, const float *k, int ks, int cout,
    const float *bias, float *y, int h) {
	const int p = (ks - 1) / 2;
	int w = 0;
	/* left border columns: taps fall outside the image */
	for (; w < p && w < W; ++w) conv_pixel(x, H, W, cin, k, ks, cout, bias, h, w, y + ((size_t)h * W + w) * cout);
	if (cout % g_block_ch == 0 && W - p > w) {
		for (; w + g_block_px <= W - p; w += g_block_px) {
			for (int ob = 0; ob < cout; ob += g_block_ch) g_block(x, H, W, cin, k, ks, cout, bias, y, h, w, ob);
		}
		for (; w < W - p; ++w) {
			for (int ob = 0; ob < cout; ob += g_block_ch) g_block1(x, H, W, cin, k, ks, cout, bias, y, h, w, ob);
		}
	}
	for (; w < W; ++w) conv_pixel(x, H, W, cin, k, ks, cout, bias, h, w, y + ((size_t)h * W + w) * cout);
}

static void conv2d_same(const float *x, int H, int W, int cin, const float *k, int ks, int cout,
    const float *bias, float *y) {
	juo_vector_bits();
#pragma omp parallel for schedule(static)
	for (int h = 0; h < H; ++h) conv_row(x, H, W, cin, k, ks, cout, bias, y, h);
}

/* keras ReLU (slope 0) / LeakyReLU(negative_slope) (models.py:24-27) */
static inline float act_fn(float v, float slope) {
	return v < 0.0f ? (slope == 0.0f ? 0.0f : v * slope) : v;
}

/* BatchNormalization at inference + optional activation (slope < 0: none), in place
 * (models.py:226-231). */
static void bn_act(float *x, size_t pixels, int c, const juo *m, const char *bn, float slope) {
	const float *g = getf(m, bn, "/gamma"), *be = getf(m, bn, "/beta");
	const float *mu = getf(m, bn, "/moving_mean"), *var = getf(m, bn, "/moving_variance");
	float sc[256], sh[256];
	for (int i = 0; i < c; ++i) {
		sc[i] = g[i] / sqrtf(var[i] + m->eps);
		sh[i] = be[i] - mu[i] * sc[i];
	}
#pragma omp parallel for schedule(static)
	for (long i = 0; i < (long)pixels; ++i) {
		float *p = x + (size_t)i * c;
		for (int o = 0; o < c; ++o) {
			float v = p[o] * sc[o] + sh[o];
			p[o] = slope < 0.0f ? v : act_fn(v, slope);
		}
	}
}

static float *conv_bn_act(const juo *m, const float *x, int H, int W, int cin, const char *conv,
    const char *bn, int *cout_out, float slope) {
	char kn[200];
	snprintf(kn, sizeof(kn), "%s/kernel", conv);
	const juo_tensor *k = get(m, kn);
	const int ks = k->dims[0], cout = k->dims[3];
	if (k->dims[2] != cin) {
		fprintf(stderr, "ju_oracle_c: %s expects %d input channels, got %d\n", conv, k->dims[2], cin);
		abort();
	}
	float *y = (float *)malloc(sizeof(float) * (size_t)H * W * cout);
	conv2d_same(x, H, W, cin, k->data, ks, cout, NULL, y);
	bn_act(y, (size_t)H * W, cout, m, bn, slope);
	*cout_out = cout;
	return y;
}

/* res_block (models.py:193-254): act(BN2(conv2(act(BN1(conv1(x))))) + x) */
static float *res_block(const juo *m, float *x, int H, int W, int c, const char *name, float slope) {
	char c1[200], b1[200], c2[200], b2[200], kn[220];
	snprintf(c1, sizeof(c1), "%s/conv_1", name);
	snprintf(b1, sizeof(b1), "%s/bn_1", name);
	snprintf(c2, sizeof(c2), "%s/conv_2", name);
	snprintf(b2, sizeof(b2), "%s/bn_2", name);
	int co;
	float *t = conv_bn_act(m, x, H, W, c, c1, b1, &co, slope);
	snprintf(kn, sizeof(kn), "%s/kernel", c2);
	const juo_tensor *k = get(m, kn);
	float *y = (float *)malloc(sizeof(float) * (size_t)H * W * c);
	conv2d_same(t, H, W, c, k->data, k->dims[0], c, NULL, y);
	free(t);
	bn_act(y, (size_t)H * W, c, m, b2, -1.0f);
	const size_t n = (size_t)H * W * c;
#pragma omp parallel for schedule(static)
	for (long i = 0; i < (long)n; ++i) y[i] = act_fn(y[i] + x[i], slope);
	free(x);
	return y;
}

/* MaxPool2D(2) (models.py:406-409) */
static float *max_pool2(float *x, int H, int W, int c) {
	const int OH = H / 2, OW = W / 2;
	float *y = (float *)malloc(sizeof(float) * (size_t)OH * OW * c);
#pragma omp parallel for schedule(static)
	for (int h = 0; h < OH; ++h) {
		for (int w = 0; w < OW; ++w) {
			for (int o = 0; o < c; ++o) {
				const float a = x[((size_t)(2 * h) * W + 2 * w) * c + o];
				const float b = x[((size_t)(2 * h) * W + 2 * w + 1) * c + o];
				const float d = x[((size_t)(2 * h + 1) * W + 2 * w) * c + o];
				const float e = x[((size_t)(2 * h + 1) * W + 2 * w + 1) * c + o];
				const float m1 = a > b ? a : b, m2 = d > e ? d : e;
				y[((size_t)h * OW + w) * c + o] = m1 > m2 ? m1 : m2;
			}
		}
	}
	free(x);
	return y;
}

/* UpscaleLayer: TF1 resize_bilinear, align_corners=False, half_pixel_centers=False
 * (keras_layers.py:46-52): src = dst / scale. Does not free x. */
static float *resize_bilinear(const float *x, int H, int W, int c, int s) {
	const int OH = H * s, OW = W * s;
	float *y = (float *)malloc(sizeof(float) * (size_t)OH * OW * c);
#pragma omp parallel for schedule(static)
	for (int oy = 0; oy < OH; ++oy) {
		const int y0 = oy / s;
		const int y1 = y0 + 1 < H ? y0 + 1 : H - 1;
		const float fy = (float)oy / (float)s - (float)y0;
		for (int ox = 0; ox < OW; ++ox) {
			const int x0 = ox / s;
			const int x1 = x0 + 1 < W ? x0 + 1 : W - 1;
			const float fx = (float)ox / (float)s - (float)x0;
			for (int o = 0; o < c; ++o) {
				const float tl = x[((size_t)y0 * W + x0) * c + o], tr = x[((size_t)y0 * W + x1) * c + o];
				const float bl = x[((size_t)y1 * W + x0) * c + o], br = x[((size_t)y1 * W + x1) * c + o];
				const float top = tl + (tr - tl) * fx;
				const float bot = bl + (br - bl) * fx;
				y[((size_t)oy * OW + ox) * c + o] = top + (bot - top) * fy;
			}
		}
	}
	return y;
}

/* Conv2DTranspose(k2, s2, "same"), kernel [2][2][cout][cin] (models.py:559-579):
 * y[2h+a, 2w+b, o] = sum_c x[h,w,c] K[a,b,o,c] (+ bias) */
static float *conv_transpose2(const float *x, int H, int W, int cin, const float *k, int cout,
    const float *bias) {
	float *y = (float *)malloc(sizeof(float) * (size_t)4 * H * W * cout);
#pragma omp parallel for schedule(static)
	for (int h = 0; h < H; ++h) {
		for (int w = 0; w < W; ++w) {
			const float *xp = x + ((size_t)h * W + w) * cin;
			for (int a = 0; a < 2; ++a) {
				for (int b = 0; b < 2; ++b) {
					float *yp = y + ((size_t)(2 * h + a) * (2 * W) + (2 * w + b)) * cout;
					for (int o = 0; o < cout; ++o) {
						const float *kr = k + ((size_t)(a * 2 + b) * cout + o) * cin;
						float acc = bias ? bias[o] : 0.0f;
						for (int c = 0; c < cin; ++c) acc += xp[c] * kr[c];
						yp[o] = acc;
					}
				}
			}
		}
	}
	return y;
}

/* ---- the graph -------------------------------------------------------- */

/* flow model on [PH][PW][3*n_in] -> head [PH][PW][32] (before depth-to-space) */
static float *flow_head(const juo *m, float *x) {
	int h = m->PH, w = m->PW, c = 3 * m->n_in, co;
	char n1[200], b1[200], n2[200], b2[200];
	if (m->arch == 0) {
		const int nb = m->n_ff / 2;
		for (int i = 0; i < 2 * nb; ++i) {
			snprintf(n1, sizeof(n1), "flow/block_%d/conv_1", i + 1);
			snprintf(b1, sizeof(b1), "flow/block_%d/bn_1", i + 1);
			snprintf(n2, sizeof(n2), "flow/block_%d/conv_2", i + 1);
			snprintf(b2, sizeof(b2), "flow/block_%d/bn_2", i + 1);
			float *a1 = conv_bn_act(m, x, h, w, c, n1, b1, &co, m->flow_slope);
			free(x);
			float *a2 = conv_bn_act(m, a1, h, w, co, n2, b2, &co, m->flow_slope);
			free(a1);
			c = co;
			if (i < nb) {
				x = max_pool2(a2, h, w, c);
				h /= 2;
				w /= 2;
			} else {
				x = resize_bilinear(a2, h, w, c, 2);
				free(a2);
				h *= 2;
				w *= 2;
			}
		}
		if (m->n_ff % 2) {
			float *a = conv_bn_act(m, x, h, w, c, "flow/conv_1", "flow/bn_1", &co, m->flow_slope);
			free(x);
			x = a;
			c = co;
		}
	} else {
		float *a = conv_bn_act(m, x, h, w, c, "flow/conv_1", "flow/bn_1", &co, m->flow_slope);
		free(x);
		x = a;
		c = co;
		for (int i = 0; i < m->res_blocks; ++i) {
			snprintf(n1, sizeof(n1), "flow/block_%d", i + 1);
			x = res_block(m, x, h, w, c, n1, m->flow_slope);
		}
	}
	const juo_tensor *k = get(m, "flow/conv_2/kernel");
	float *y = (float *)malloc(sizeof(float) * (size_t)h * w * 32);
	conv2d_same(x, h, w, c, k->data, k->dims[0], 32, get(m, "flow/conv_2/bias")->data, y);
	free(x);
	return y;
}

void *juo_create(const void *blob, size_t size) {
	static const char magic[8] = {'J', 'U', 'P', 'W', 'G', 'T', 0, 1};
	if (size < 128 || memcmp(blob, magic, 8) != 0) return NULL;
	juo *m = (juo *)calloc(1, sizeof(juo));
	m->blob = (unsigned char *)malloc(size);
	memcpy(m->blob, blob, size);
	const unsigned char *b = m->blob;
	const uint32_t hdr = rd32(b + 12);
	m->H = (int)rd32(b + 16);
	m->W = (int)rd32(b + 20);
	m->n_in = (int)rd32(b + 28);
	m->arch = (int)rd32(b + 32);
	m->pad = (int)rd32(b + 36);
	m->gen_filters = (int)rd32(b + 44);
	m->gen_blocks = (int)rd32(b + 48);
	m->res_filters = (int)rd32(b + 52);
	m->res_blocks = (int)rd32(b + 56);
	m->n_ff = (int)rd32(b + 60);
	for (int i = 0; i < m->n_ff && i < 8; ++i) m->ff[i] = (int)rd32(b + 64 + 4 * i);
	memcpy(&m->eps, b + 96, 4);
	{ /* header word 116: activation codes (0 relu, 1 lrelu), 120 / 124: negative slopes */
		const uint32_t acts = rd32(b + 116);
		m->flow_slope = m->gen_slope = 0.0f;
		if ((acts & 0xff) == 1) memcpy(&m->flow_slope, b + 120, 4);
		if (((acts >> 8) & 0xff) == 1) memcpy(&m->gen_slope, b + 124, 4);
	}
	m->n_tensors = (int)rd32(b + 104);
	m->normalize = rd32(b + 40) != 0;
	if (rd32(b + 108) != 0 || m->n_tensors > MAX_TENSORS) { /* temporal filter: unsupported */
		free(m->blob);
		free(m);
		return NULL;
	}
	/* models.py:735-744 */
	m->PH = m->pad ? (m->H + m->pad - 1) / m->pad * m->pad : m->H;
	m->PW = m->pad ? (m->W + m->pad - 1) / m->pad * m->pad : m->W;
	for (int i = 0; i < m->n_tensors; ++i) {
		const unsigned char *e = b + hdr + (size_t)i * 128;
		memcpy(m->t[i].name, e, 92);
		m->t[i].name[92] = 0;
		m->t[i].ndim = (int)rd32(e + 92);
		for (int d = 0; d < 4; ++d) m->t[i].dims[d] = (int)rd32(e + 96 + 4 * d);
		m->t[i].data = (const float *)(b + rd64(e + 112));
		m->t[i].count = (size_t)rd64(e + 120);
	}
	const size_t hr = (size_t)16 * m->H * m->W * 3;
	m->pre_gen = (float *)calloc(hr, sizeof(float));
	m->output_raw = (float *)calloc(hr, sizeof(float));
	for (int i = 0; i < m->n_in - 1; ++i) {
		m->last[i] = (float *)calloc((size_t)m->PH * m->PW * 3, sizeof(float));
	}
	return m;
}

void juo_destroy(void *h) {
	juo *m = (juo *)h;
	if (!m) return;
	for (int i = 0; i < 8; ++i) free(m->last[i]);
	free(m->pre_gen);
	free(m->output_raw);
	free(m->blob);
	free(m);
}

void juo_reset(void *h) {
	juo *m = (juo *)h;
	memset(m->pre_gen, 0, sizeof(float) * 16 * m->H * m->W * 3);
	for (int i = 0; i < m->n_in - 1; ++i) memset(m->last[i], 0, sizeof(float) * m->PH * m->PW * 3);
}

const float *juo_output_raw(void *h) {
	return ((juo *)h)->output_raw;
}

/* One recurrent step on a dense BGRX frame; writes a dense BGRX frame (X = 0).
 * Loop semantics: scripts/inference/onnx/inference.py:72-94. */
int juo_run(void *hdl, const uint8_t *frame, uint8_t *out) {
	juo *m = (juo *)hdl;
	const int H = m->H, W = m->W, PH = m->PH, PW = m->PW, HH = 4 * H, WW = 4 * W;
	const int pt = (PH - H) / 2, pl = (PW - W) / 2;
	/* PreprocessLayer (keras_layers.py:208); X byte ignored */
	float *cur = (float *)malloc(sizeof(float) * (size_t)H * W * 3);
	for (size_t i = 0; i < (size_t)H * W; ++i) {
		for (int c = 0; c < 3; ++c) cur[i * 3 + c] = (float)frame[i * 4 + c] / 255.0f - 0.5f;
	}
	/* normalize_brightness (models.py:772-779): the frame's luma mean, reduce_mean(cur * BGR_LUMA * 3) over all
	 * H x W x 3 elements = the luma-weighted mean over the pixels; taken off the FLOW net's input only */
	float bright = 0.0f;
	if (m->normalize) {
		static const double luma[3] = {0.114, 0.587, 0.2989};
		double acc = 0.0;
		for (size_t i = 0; i < (size_t)H * W; ++i) {
			for (int c = 0; c < 3; ++c) acc += (double)cur[i * 3 + c] * luma[c];
		}
		bright = (float)(acc / ((double)H * W));
	}
	/* ZeroPadding2D (models.py:780-789) */
	float *cur_pad = (float *)calloc((size_t)PH * PW * 3, sizeof(float));
	for (int y = 0; y < H; ++y) {
		float *dst = cur_pad + ((size_t)(y + pt) * PW + pl) * 3;
		const float *src = cur + (size_t)y * W * 3;
		for (int i = 0; i < W * 3; ++i) dst[i] = src[i] - bright;
	}
	/* concat [cur_pad] + last_frames (models.py:373-375, 790) */
	const int fc = 3 * m->n_in;
	float *fin = (float *)malloc(sizeof(float) * (size_t)PH * PW * fc);
	for (size_t i = 0; i < (size_t)PH * PW; ++i) {
		memcpy(fin + i * fc, cur_pad + i * 3, sizeof(float) * 3);
		for (int j = 0; j < m->n_in - 1; ++j) {
			memcpy(fin + i * fc + 3 * (j + 1), m->last[j] + i * 3, sizeof(float) * 3);
		}
	}
	float *head = flow_head(m, fin); /* frees fin */
	/* depth_to_space(4) + unpad crop (keras_layers.py:175; models.py:791-798) fused into
	 * the warp's flow lookup; dense_image_warp (tfa/dense_image_warp.py:232-245, 116-171) */
	float *pre_warp = (float *)malloc(sizeof(float) * (size_t)HH * WW * 3);
#pragma omp parallel for schedule(static)
	for (int Y = 0; Y < HH; ++Y) {
		for (int X = 0; X < WW; ++X) {
			const float *f = head + ((size_t)(Y / 4 + pt) * PW + (X / 4 + pl)) * 32 + ((Y % 4) * 4 + (X % 4)) * 2;
			const float qy = (float)Y - f[0], qx = (float)X - f[1];
			float fy = floorf(qy), fx = floorf(qx);
			fy = fy < 0.0f ? 0.0f : (fy > (float)(HH - 2) ? (float)(HH - 2) : fy);
			fx = fx < 0.0f ? 0.0f : (fx > (float)(WW - 2) ? (float)(WW - 2) : fx);
			float ay = qy - fy, ax = qx - fx;
			ay = ay < 0.0f ? 0.0f : (ay > 1.0f ? 1.0f : ay);
			ax = ax < 0.0f ? 0.0f : (ax > 1.0f ? 1.0f : ax);
			const int y0 = (int)fy, x0 = (int)fx;
			for (int c = 0; c < 3; ++c) {
				const float tl = m->pre_gen[((size_t)y0 * WW + x0) * 3 + c];
				const float tr = m->pre_gen[((size_t)y0 * WW + x0 + 1) * 3 + c];
				const float bl = m->pre_gen[((size_t)(y0 + 1) * WW + x0) * 3 + c];
				const float br = m->pre_gen[((size_t)(y0 + 1) * WW + x0 + 1) * 3 + c];
				const float top = ax * (tr - tl) + tl;
				const float bot = ax * (br - bl) + bl;
				pre_warp[((size_t)Y * WW + X) * 3 + c] = ay * (bot - top) + top + bright; /* models.py:802-803 */
			}
		}
	}
	free(head);
	/* generator input: concat [images, space_to_depth(pre_warp, 4)] (models.py:523-530) */
	float *x = (float *)malloc(sizeof(float) * (size_t)H * W * 51);
	for (int h = 0; h < H; ++h) {
		for (int w = 0; w < W; ++w) {
			float *p = x + ((size_t)h * W + w) * 51;
			memcpy(p, cur + ((size_t)h * W + w) * 3, sizeof(float) * 3);
			for (int i = 0; i < 4; ++i) {
				for (int j = 0; j < 4; ++j) {
					memcpy(p + 3 + (i * 4 + j) * 3, pre_warp + ((size_t)(4 * h + i) * WW + 4 * w + j) * 3,
					    sizeof(float) * 3);
				}
			}
		}
	}
	free(pre_warp);
	int c;
	float *g = conv_bn_act(m, x, H, W, 51, "generator/conv_1", "generator/bn_1", &c, m->gen_slope);
	free(x);
	char name[200];
	for (int i = 0; i < m->gen_blocks; ++i) {
		snprintf(name, sizeof(name), "generator/block_%d", i + 1);
		g = res_block(m, g, H, W, c, name, m->gen_slope);
	}
	const juo_tensor *k1 = get(m, "generator/conv_trans_1/kernel");
	float *t1 = conv_transpose2(g, H, W, c, k1->data, k1->dims[2], NULL);
	free(g);
	bn_act(t1, (size_t)4 * H * W, k1->dims[2], m, "generator/bn_2", m->gen_slope);
	const juo_tensor *k2 = get(m, "generator/conv_trans_2/kernel");
	float *t2 = conv_transpose2(t1, 2 * H, 2 * W, k1->dims[2], k2->data, 3,
	    get(m, "generator/conv_trans_2/bias")->data);
	free(t1);
	float *up = resize_bilinear(cur, H, W, 3, 4); /* models.py:584-587 */
	const size_t n = (size_t)HH * WW * 3;
#pragma omp parallel for schedule(static)
	for (long i = 0; i < (long)n; ++i) {
		float v = tanhf(t2[i]) + up[i];            /* a_3 + add, models.py:580-590 */
		v = v < -0.5f ? -0.5f : (v > 0.5f ? 0.5f : v); /* ClipLayer */
		m->output_raw[i] = v;
	}
	free(t2);
	free(up);
	/* PostprocessLayer + truncating cast; X = 0 (keras_layers.py:227-230,
	 * core/src/cuda_convert.cc.cu:39-45, 76-81) */
	for (size_t i = 0; i < (size_t)HH * WW; ++i) {
		for (int ch = 0; ch < 3; ++ch) out[i * 4 + ch] = (uint8_t)((m->output_raw[i * 3 + ch] + 0.5f) * 255.0f);
		out[i * 4 + 3] = 0;
	}
	/* output_raw leaves the graph with the brightness taken off again (models.py:809-810) */
	if (m->normalize) {
#pragma omp parallel for schedule(static)
		for (long i = 0; i < (long)n; ++i) m->output_raw[i] -= bright;
	}
	/* state update (models.py:821-823) */
	memcpy(m->pre_gen, m->output_raw, sizeof(float) * n);
	if (m->n_in > 1) {
		float *oldest = m->last[m->n_in - 2];
		for (int j = m->n_in - 2; j > 0; --j) m->last[j] = m->last[j - 1];
		m->last[0] = oldest;
		memcpy(m->last[0], cur_pad, sizeof(float) * (size_t)PH * PW * 3);
	}
	free(cur_pad);
	free(cur);
	return 0;
}
