/*
 * joshupscale_amd.h -- C ABI of the MI355X-native JoshUpscale runtime.
 *
 * This is the drop-in boundary for the reference's core/ runtime: plain C,
 * plain pointers and sizes, no C++ or torch types.  Each entry point states the
 * reference interface it replaces (paths relative to the reference repository).
 * The C++ surface the AviSynth/OBS plugins compile against
 * (include/JoshUpscale/core.h) is a thin shim over these functions; so are the
 * Python (ctypes) bindings in joshupscale_amd/runtime.py.
 *
 * Threading: a ju_runtime is not thread-safe (one internal HIP stream; the
 * reference declares MT_SERIALIZED, avisynth_plugin/src/main.cc:176-178).
 * Several runtimes, also on different devices, may coexist.  ju_last_error()
 * is thread-local.
 *
 * Every function returning int returns JU_OK (0) or a JU_ERR_* code; the
 * message is then available from ju_last_error() on the calling thread.
 */
#ifndef JOSHUPSCALE_AMD_H_
#define JOSHUPSCALE_AMD_H_

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define JU_API __attribute__((visibility("default")))
#else
#define JU_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ju_runtime ju_runtime;

enum {
	JU_OK = 0,
	JU_ERR_INVALID_ARGUMENT = 1, /* std::invalid_argument in the reference */
	JU_ERR_IO = 2,               /* std::ios_base::failure (model file) */
	JU_ERR_DEVICE = 3,           /* HIP error (CudaException, core/include/JoshUpscale/core/cuda.h:24-32) */
	JU_ERR_UNSUPPORTED = 4,
	JU_ERR_INTERNAL = 5
};

/* DataLocation of core/public/JoshUpscale/core.h:30.  JU_LOC_DEVICE is the
 * reference's CUDA location reinterpreted as "HIP device pointer". */
enum { JU_LOC_CPU = 0, JU_LOC_DEVICE = 1, JU_LOC_GRAPHICS_RESOURCE = 2 };

/* Compute precision of MFMA operands / stored activations.  JU_DTYPE_FP8: the 64->64
 * residual-block convolutions of the generator run on OCP e4m3 operands (block-scaled
 * MFMA; the counterpart of the reference's TensorRT INT8 engines,
 * scripts/inference/tensorrt/quantize_int8.py:140-209), every other layer and the
 * residual stream stay fp16. */
enum { JU_DTYPE_DEFAULT = -1, JU_DTYPE_F16 = 0, JU_DTYPE_BF16 = 1, JU_DTYPE_FP8 = 2 };

/* LogLevel of core/public/JoshUpscale/core.h:21. */
enum { JU_LOG_INFO = 0, JU_LOG_WARNING = 1, JU_LOG_ERROR = 2 };

/* Mirrors struct Image (core/public/JoshUpscale/core.h:32-38) field for field:
 * 4 bytes per pixel, byte order B,G,R,X (X ignored on input, written 0 on
 * output); stride in bytes, may be negative (bottom-up frames, ptr = first
 * logical row); width/height in pixels. */
typedef struct ju_image {
	void *ptr;
	uint8_t location;
	ptrdiff_t stride;
	size_t width;
	size_t height;
} ju_image;

/* Replaces createRuntime(int deviceId, const std::filesystem::path &modelPath)
 * (core/public/JoshUpscale/core.h:91-92, core/src/core.cc:153-175, 197-199):
 * reads the whole model file, selects the device for the duration of the call,
 * builds the engine.  The file is this runtime's .jupw container
 * (joshupscale_amd/model_file.py), not a TensorRT engine. */
JU_API int ju_create(int device_id, const char *model_path, ju_runtime **out_runtime);

/* Same, from model bytes already in memory (what TensorRTBackend's constructor
 * takes: core/src/tensorrt_backend.cc:117).  dtype: JU_DTYPE_*.  Used by the
 * multi-GPU launcher after the RCCL weight broadcast. */
JU_API int ju_create_from_memory(int device_id, const void *model_bytes, size_t model_size,
    int dtype, ju_runtime **out_runtime);

/* Parses and checks a model container WITHOUT touching a device: header ranges,
 * tensor table bounds, presence and shape of every layer's variables, the channel
 * chain of the graph, BatchNorm folding.  The same code runs first inside
 * ju_create*; this entry point lets a caller (or a CI job on a box without a GPU)
 * reject a bad or hostile file early.  No reference counterpart: TensorRT's
 * deserializeCudaEngine is the validator there (core/src/tensorrt_backend.cc:145-148).
 * JU_OK, or JU_ERR_INVALID_ARGUMENT with the reason in ju_last_error(). */
JU_API int ju_validate_model(const void *model_bytes, size_t model_size);

/* Replaces Runtime::~Runtime via delete (core.h:65-66). NULL is a no-op. */
JU_API void ju_destroy(ju_runtime *runtime);

/* Replaces Runtime::processImage(const Image&, const Image&) (core.h:68-69,
 * core/src/core.cc:177-189, core/src/tensorrt_backend.cc:270-278): stage-in,
 * one recurrent step, stage-out, stream synchronise, state ping-pong.
 * Unlike the reference (assert only, core.cc:179-182) wrong sizes are an
 * error (JU_ERR_INVALID_ARGUMENT). */
JU_API int ju_process(ju_runtime *runtime, const ju_image *input, const ju_image *output);

/* Frame look-ahead (no reference counterpart; the reference's callers hand over one frame at a time,
 * avisynth_plugin/src/main.cc:113-144): `count` CONSECUTIVE frames of the stream in one synchronous call.
 * outputs[i] receives exactly the bytes ju_process(inputs[i], outputs[i]) called in order would have written,
 * and the recurrent state afterwards is the same -- but ALL inputs must hold their pixels when the call is made
 * (an input that overlaps the OUTPUT of an earlier frame of the call -- to be read after that write, frame by frame --
 * simply starts a new pass, and so does an output that overlaps an earlier frame's INPUT).
 * The flow net reads LR frames only, never the HR state, so the runtime computes the flow fields of up to 8 frames in
 * ONE pass of the flow net's launches, which fill the chip where one frame's do not (-40 % flow time per frame at
 * 480x270); warp, tower and tail stay strictly frame by frame.  JU_LOC_DEVICE frames are read and written in place;
 * host frames (JU_LOC_CPU) ride in the same passes -- every input uploaded up front, each output copied out while the
 * next frame's kernels run (pageable memory; nothing of the caller's is page-locked).  Frames a pass cannot take (GL
 * resources, a model without the one-launch flow plan) simply run as ju_process does.  For callers that can read
 * ahead: a file transcoder, an AviSynth filter fetching child frames n .. n+7.
 * ju_set_lookahead caps the frames per pass (1 = off); its default is 8, or JU_LOOKAHEAD=<1..8> at creation. */
JU_API int ju_process_batch(ju_runtime *runtime, const ju_image *inputs, const ju_image *outputs, int count);
/* Frames per look-ahead pass of ju_process_batch for THIS runtime, 1 (every frame as ju_process does) .. 8; values
 * outside are clamped.  Passes registered with ju_prepare_batch that are longer than the new cap are forgotten; RAISING
 * the cap after passes have run or been registered re-allocates the passes' tensors, and every pass graph is then
 * captured again at its next use (set the cap once, before ju_prepare_batch).  The
 * setter is what a host application uses; the JU_LOOKAHEAD environment variable only sets the default of runtimes
 * created afterwards (one process, several filters: each sets its own). */
JU_API int ju_set_lookahead(ju_runtime *runtime, int frames);
/* What ju_prepare_frames is to ju_process: registers a tuple of 2 .. cap (ju_set_lookahead) frame buffers -- device or host -- the
 * caller is going to hand to ju_process_batch as one pass; its hipGraphs (one per binding set) are captured now,
 * nothing executes.  Unregistered tuples are captured at their second use.  *captured (optional) = graphs captured
 * by this call; 0 for a tuple that will not run as one pass. */
JU_API int ju_prepare_batch(ju_runtime *runtime, const ju_image *inputs, const ju_image *outputs, int count, int *captured);

/* Asynchronous form for JU_LOC_DEVICE images: enqueues the same work on the
 * runtime's stream and returns; ju_synchronize() waits.  Frames are still
 * strictly ordered (the recurrence is carried by stream order). */
JU_API int ju_enqueue(ju_runtime *runtime, const ju_image *input, const ju_image *output);
JU_API int ju_synchronize(ju_runtime *runtime);

/* Registers a pair of JU_LOC_DEVICE frame buffers the caller is going to pass to ju_process /
 * ju_enqueue: the hipGraphs of the pair (one per binding set) are captured NOW, so that no
 * later call captures anything -- the reference captures its two graphs in the constructor
 * (core/src/tensorrt_backend.cc:257-263), never inside process (:270-278).  Nothing executes
 * and the buffers are not read or written.  Callers reuse a handful of buffers (OBS: one
 * texture pair, obs_plugin/src/filter.cc:242-279); a caller that does not register still gets
 * a graph from the second use of a pair on.  JU_LOC_CPU / graphics-resource images need
 * nothing (their frames go through the staging buffers whose graphs exist from ju_create on):
 * the call checks the sizes and returns.  *captured (may be NULL) receives the number of
 * graphs captured by this call: 2 for a new device pair, 0 otherwise.  Up to 256 pairs stay
 * registered; beyond that the pair used least recently is forgotten with its graphs. */
JU_API int ju_prepare_frames(ju_runtime *runtime, const ju_image *input, const ju_image *output, int *captured);

/* Replaces Runtime::getInputWidth/Height, getOutputWidth/Height (core.h:71-82). */
JU_API int ju_get_size(const ju_runtime *runtime, size_t *input_width, size_t *input_height,
    size_t *output_width, size_t *output_height);

/* Zeroes the recurrent state; equivalent to destroying and recreating the
 * runtime as the OBS filter does on a model switch (obs_plugin/src/filter.cc:146-151). */
JU_API int ju_reset(ju_runtime *runtime);

/* Replaces getExceptionString() (core.h:94): message of the last failed call on
 * this thread ("" if none). The pointer stays valid until the next failing
 * call on the same thread. */
JU_API const char *ju_last_error(void);

/* Replaces setLogSink(LogSink*) (core.h:23-28). callback NULL restores the
 * default sink (stderr, warnings and errors only unless JU_VERBOSE=1). */
typedef void (*ju_log_callback)(const char *tag, int level, const char *message, void *user);
JU_API void ju_set_log_callback(ju_log_callback callback, void *user);

/* Replaces getGLDeviceIndex() (core.h:60, core/src/core.cc:140-149): the HIP device that
 * drives the calling thread's current OpenGL context (hipGLGetDevices). */
JU_API int ju_get_gl_device_index(int *out_device);

/* Replaces getGLImage(image, type) (core.h:61-62, core.cc:92-138): registers an OpenGL
 * 2-D texture (RGBA8 / BGRX, the caller's GL context current) with the HIP runtime
 * (hipGraphicsGLRegisterImage; type 0 = input, read only; 1 = output, write discard) and
 * describes it as a JU_LOC_GRAPHICS_RESOURCE image whose width / height are the
 * texture's.  ju_process maps it, copies texture array <-> staging buffer and unmaps it
 * (core/include/JoshUpscale/core/cuda.h:310-349, core/src/cuda_convert.cc.cu:380-397,
 * 419-436).  Release with ju_release_gl_image (the reference's ~GLResourceImage). */
JU_API int ju_get_gl_image(uint32_t gl_texture, int type, ju_image *out_image);
JU_API void ju_release_gl_image(ju_image *image);

/* ---- multi-GPU start-up (BASELINE.json config 4; no reference counterpart: the reference
 * has no distributed code).  N GPUs = N independent streams, one process per GPU; the ONLY
 * collective is the broadcast of the model container from rank 0, so that one rank reads
 * the file.  RCCL (librccl, opened at first use) over xGMI.  The launcher carries the
 * JU_COMM_ID_BYTES id from the rank that called ju_comm_unique_id to the others
 * (bench.py: through torch.distributed's rendezvous). ---------------------------------- */
typedef struct ju_comm ju_comm;
#define JU_COMM_ID_BYTES 128
JU_API int ju_comm_unique_id(void *id_out /* JU_COMM_ID_BYTES */);
JU_API int ju_comm_create(const void *id, int rank, int world_size, int device_id, ju_comm **out_comm);
/* ncclBroadcast of `size` bytes (uint8) from `root`'s buffer into every rank's buffer. */
JU_API int ju_comm_broadcast(ju_comm *comm, void *bytes, size_t size, int root);
/* In-place maximum over ranks of one double (the bench's max-over-ranks elapsed time). */
JU_API int ju_comm_allreduce_max(ju_comm *comm, double *value);
/* ncclCommCount: the number of ranks this communicator really spans. */
JU_API int ju_comm_count(const ju_comm *comm, int *count);
JU_API void ju_comm_destroy(ju_comm *comm);

/* ---- introspection (no reference counterpart).  The test and measurement hooks (ju_debug_*,
 * ju_read_tensor, ju_time_steps) are NOT part of this library: they are declared in
 * joshupscale_amd_test.h and exported only by libJoshUpscale_test.so (built with -DJU_TEST_HOOKS);
 * the reference exports nothing but its JOSHUPSCALE_EXPORT symbols (core/CMakeLists.txt:29-36). -- */

/* Compute dtype actually in use (JU_DTYPE_F16 / JU_DTYPE_BF16 / JU_DTYPE_FP8). */
JU_API int ju_get_dtype(const ju_runtime *runtime);

/* How the runtime has been executing: "graph_replays" / "eager_runs" (per-frame programs
 * submitted as one hipGraph replay / as individual launches so far), "graph_captures"
 * (graphs captured INSIDE ju_process / ju_enqueue so far; captures by ju_prepare_frames:
 * "prepared_captures"), "registered_pairs", "direct_graphs"
 * (graphs cached for JU_LOC_DEVICE frame tuples), "resident_tower" / "resident_flow"
 * (1 when the one-launch tower kernel is in use), "launches_per_frame", "tower_variant". */
JU_API int ju_get_stat(const ju_runtime *runtime, const char *key, double *value);

/* Library version string, e.g. "joshupscale-amd 0.1 (gfx950)". */
JU_API const char *ju_version(void);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* JOSHUPSCALE_AMD_H_ */
