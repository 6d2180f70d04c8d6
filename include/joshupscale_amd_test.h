/* Test and measurement hooks of the MI355X engine.  NOT part of the product ABI: the shipped
 * libJoshUpscale.so exports none of these (tests/test_c_abi.py checks `nm -D`); they exist only in
 * libJoshUpscale_test.so, the same objects linked with c_api.cpp / graphics.cpp compiled under
 * -DJU_TEST_HOOKS (Makefile).  The reference hides everything that is not JOSHUPSCALE_EXPORT
 * (core/CMakeLists.txt:29-36): a plugin host must not be able to flip engine behaviour.
 * Users: tests/, bench.py (ju_time_steps for the roofline's in-frame kernel time), tools/. */
#ifndef JOSHUPSCALE_AMD_TEST_H_
#define JOSHUPSCALE_AMD_TEST_H_

#include "joshupscale_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test double for the graphics path (no GL context exists on a headless GPU box): texture
 * ids defined here resolve to pitched device buffers; device_ptr NULL removes the double
 * again.  Counters: resources registered / currently mapped / map and unmap calls. */
JU_API int ju_debug_fake_gl_texture(uint32_t gl_texture, void *device_ptr, size_t pitch, size_t width,
    size_t height, int bytes_per_pixel);
JU_API void ju_debug_fake_gl_counters(int *registered, int *mapped, int *maps, int *unmaps);

/* Copies a named internal tensor to host memory as float32.  *count receives the
 * element count; dst may be NULL to query it.  Names: "state" (last output_raw,
 * f16 [4H][4W][4]), "flow" (f16 [PH][PW][32], the flow head before depth-to-space), "flow_in", "gen_in", "trunk",
 * "tail_y", and the per-layer flow activations.  "flow" and the flow activations are the PER-FRAME tensors: a look-ahead pass
 * (ju_process_batch) computes its flow fields in tensors of its own and does not update them; "state", "flow_in" and "gen_in"
 * are what the last frame of a pass left, as after ju_process. */
JU_API int ju_read_tensor(ju_runtime *runtime, const char *name, float *dst, size_t capacity,
    size_t *count);

/* Average device time in milliseconds of ONE kernel launch among the per-frame
 * steps tagged `tag` ("tower" = the 3x3 64->64 convolutions of the generator's
 * residual blocks, "flow", "warp", "gen_head", "tail", "pack", "" = all),
 * measured with HIP events on the runtime's own stream over `iters`
 * repetitions.  *launches = kernel launches per repetition, *flops = their
 * algorithmic FLOPs (2*MAC) per repetition.  "tag#k": only the k-th launch of the tag.
 * "tag@frame" (also "tag#k@frame"): the tagged launches timed INSIDE whole frames -- every step
 * of the frame runs, HIP events bracket the tagged launches -- i.e. the kernel in the clock and
 * cache context of the real workload (what a kernel trace of the benchmark averages).  Timing
 * overwrites scratch tensors and the recurrent state: the state is reset (as by ju_reset)
 * before the call returns. */
JU_API int ju_time_steps(ju_runtime *runtime, const char *tag, int iters, double *ms_per_launch,
    int *launches, double *flops);

/* Developer switches (timing ablations and fault injection; never needed by a
 * caller).  Keys: "tower_variant" (0 = product kernel, 4 = phase profile, 5 = per-layer
 * output maxima for quantisation calibration, 8 = the resident tower's plain schedule:
 * same bytes, tests compare it with the product's); "resident_fault" n
 * (launch the resident tower n workgroups short: tests the fallback). */
JU_API int ju_debug_set(const char *key, int value);

/* The loader's e4m3 quantiser (round to nearest even, saturating at +-448), exposed so
 * that the CPU tests can pin it against the oracle's restatement.  No device needed. */
JU_API int ju_debug_e4m3(const float *values, unsigned char *codes, size_t count);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* JOSHUPSCALE_AMD_TEST_H_ */
