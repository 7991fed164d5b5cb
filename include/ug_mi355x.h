/*
 * ug_mi355x.h -- C ABI of libug_mi355x.so, the MI355X (gfx950 / CDNA4) kernel library for
 * UltraGrid's per-frame pixel-format-conversion + block-compression hot path.
 *
 * This header is the drop-in boundary.  It replaces, for this path,
 *   - cuda_dxt/cuda_dxt.h:30-89      (cuda_{rgb,yuv}_to_dxt{1,6}, cuda_yuv422_to_yuv444)
 *   - src/cuda_wrapper.h:50-76       (device select / alloc / memcpy / last-error shim)
 *   - the per-line CPU decoder_t loop every compress module runs before upload
 *     (src/pixfmt_conv.h:87-88, src/video_compress/cuda_dxt.cpp:206-220,
 *      src/video_compress/dxt_glsl.cpp:277-289, src/video_compress/gpujpeg.cpp:592-608)
 *   - the packed->planar whole-buffer converters src/to_planar.h:53-74
 *   - the FDCT+quantise stage UltraGrid gets from libgpujpeg
 *     (src/video_compress/gpujpeg.cpp:617-631)
 *
 * Conventions (same as cuda_dxt.h): plain C, no C++ types, no ownership transfer (the
 * caller owns every buffer), every function returns 0 on success and a negative
 * UG_HIP_E* code on failure and never throws; distinct streams may be driven from
 * distinct threads concurrently.  Kernels are ASYNCHRONOUS on `stream` (unlike
 * cuda_dxt.cu:759, which synchronises after every launch) -- call ug_hip_stream_sync()
 * or order a D2H copy on the same stream.  `stream` == NULL is the device's null stream.
 *
 * Image geometry: `width`/`height` in pixels; a NEGATIVE height means the source image is
 * read bottom-up (vertical mirror), exactly as cuda_dxt.h:38-39.  `src_pitch` is the
 * source line stride in bytes; 0 selects the tightly packed UltraGrid line size
 * (vc_get_linesize, src/video_codec.c:507-521).
 *
 * Argument ranges: width and |height| from 1 to 65536 (UltraGrid's largest mode is 8K) and at most INT_MAX bytes per frame or plane
 * (pitch x lines).  Anything outside -- sizes that are not a picture, or whose byte counts would leave int / size_t range somewhere -- is
 * refused with UG_HIP_EINVAL before any device call (cuda_dxt.cu:745-746 returns -1 for a bad size); no entry point returns success, a
 * wrapped value or a runtime error for such input (tests/test_abi.py::test_absurd_geometry_is_refused, every entry point of this header).
 */
#ifndef UG_MI355X_H
#define UG_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UG_HIP_ABI_VERSION 5 /* 5: ug_hip_memcpy_2d_async, ug_hip_download_2d_ordered_ex, ug_hip_event_* / ug_hip_stream_wait_event, ug_hip_jpeg_encoder_create_ex, ug_hip_jpeg_colour_* (additions), and the DXT encoder / decoders take ANY frame size (they refused sizes
                              * that are not multiples of 4; ug_hip_dxt_size rounds up to whole blocks as dxt_get_size does); 4: ug_hip_{upload,download}_ordered_ex (additions only); 2: tie-rule option (UG_DXT_TIES_*), default = ties to even; *_ex / batched entry points; 3: NUMA placement, de-interlace (additions), and ONE
                              * change of behaviour: ug_hip_jpeg_encoder_encode_batch with frames > 1 reports a stream that does not fit its slice through
                              * out_len[f] > out_capacity and returns success for the call (it used to fail the whole call with UG_HIP_EINVAL) */

/* error codes (cuda_dxt.cu:745-746,759 uses -1 bad size/alignment, -3 runtime failure) */
#define UG_HIP_SUCCESS      0
#define UG_HIP_EINVAL     (-1) /* bad size / alignment / NULL pointer */
#define UG_HIP_EUNSUPP    (-2) /* no kernel for this format pair */
#define UG_HIP_ERUNTIME   (-3) /* HIP runtime error: see ug_hip_last_error_string() */

typedef void *ug_hip_stream_t; /* == hipStream_t; replaces cuda_wrapper_stream_t */

/* Pixel formats (own numbering; UltraGrid codec_t values are mapped by the module shim,
 * ultragrid_amd/module/ug_codec_map.h). */
typedef enum {
        UG_PF_NONE = 0,
        UG_PF_RGBA = 1,     /* 8-bit R,G,B,A bytes (shifts configurable on output) */
        UG_PF_UYVY = 2,     /* 8-bit 4:2:2  U Y0 V Y1 */
        UG_PF_YUYV = 3,     /* 8-bit 4:2:2  Y0 U Y1 V */
        UG_PF_RGB  = 4,     /* 8-bit R,G,B */
        UG_PF_BGR  = 5,     /* 8-bit B,G,R */
        UG_PF_V210 = 6,     /* 10-bit 4:2:2, 6 px / 16 B, lines padded to 128 B */
        UG_PF_RG48 = 7,     /* 16-bit little-endian R,G,B */
        UG_PF_YUV444 = 8,   /* packed 8-bit Y,U,V triplets (output of *_yuv422_to_yuv444) */
        UG_PF_UYVY_RAW = 9, /* UYVY fed to the encoder WITHOUT colour conversion (DXT1_YUV) */
        UG_PF_I420 = 10,    /* planar 4:2:0: Y, U, V planes back to back (JPEG encoder input only) */
        /* the other codecs of decoders[] (pixfmt_conv.c:3041-3103), ug_hip_pixfmt_convert only: */
        UG_PF_R10K = 11,    /* 10-bit RGB, 4 B / px big-endian, lines padded to 64 px (types.h R10k) */
        UG_PF_R12L = 12,    /* 12-bit RGB, 8 px / 36 B little-endian bit stream (R12L) */
        UG_PF_Y216 = 13,    /* 16-bit 4:2:2  Y0 Cb Y1 Cr */
        UG_PF_Y416 = 14,    /* 16-bit 4:4:4:4  U Y V A */
        UG_PF_VUYA = 15,    /* 8-bit 4:4:4:4  V U Y A */
        UG_PF_DVS10 = 16,   /* 10-bit 4:2:2 of DVS cards, 6 px / 16 B like v210 */
} ug_pixfmt_t;

typedef enum {
        UG_DXT1       = 1, /* 8 B / 4x4 block, output size w*h/2 (dxt_util.h:59-67) */
        UG_DXT1_YUV   = 2, /* DXT1 blocks holding Y,Cb,Cr (-c RTDXT:DXT1_YUV, types.h DXT1_YUV): encode takes UG_PF_UYVY only
                            * (== UG_PF_UYVY_RAW -> UG_DXT1), decode applies display_dxt1_yuv_fp.glsl */
        UG_DXT5_YCOCG = 6, /* "DXT6": 16 B / block, output size w*h */
} ug_dxt_t;

/* ------------------------------------------------------------------------------------
 * Runtime shim (replaces src/cuda_wrapper.h:50-76)
 * ---------------------------------------------------------------------------------- */
int         ug_hip_abi_version(void);
int         ug_hip_device_count(int *count);
/* 1 if ptr is device memory of this process (a device-resident video_frame, types.h:295-298 mem_location; the reference's
 * GPUJPEG module takes such frames without the upload, gpujpeg.cpp:617-622), 0 for host / unknown pointers. Never fails. */
int         ug_hip_pointer_is_device(const void *ptr);
/* index of the device that owns ptr, -1 for host / unknown pointers: a module state bound to device d reads a device-resident frame in
 * place only when this returns d (a frame that lives on another GPU is copied over first). Never fails. */
int         ug_hip_pointer_device(const void *ptr);
int         ug_hip_set_device(int index);                              /* cuda_wrapper_set_device */
int         ug_hip_malloc(void **buffer, size_t size);                 /* cuda_wrapper_malloc */
int         ug_hip_free(void *buffer);                                 /* cuda_wrapper_free */
int         ug_hip_malloc_host(void **buffer, size_t size);            /* cuda_wrapper_malloc_host (pinned) */
int         ug_hip_free_host(void *buffer);                            /* cuda_wrapper_free_host */
#define UG_HIP_MEMCPY_HOST_TO_DEVICE 0
#define UG_HIP_MEMCPY_DEVICE_TO_HOST 1
#define UG_HIP_MEMCPY_DEVICE_TO_DEVICE 2
int         ug_hip_memcpy(void *dst, const void *src, size_t count, int kind);       /* cuda_wrapper_memcpy */
int         ug_hip_memcpy_async(void *dst, const void *src, size_t count, int kind, ug_hip_stream_t stream);
/* `rows` lines of `width_bytes` bytes, `spitch` / `dpitch` bytes from one line to the next: ONE copy for a picture whose pitch differs from its
 * line size (the receivers' display pitch: src/video_decompress/dxt_glsl.c:163-186, gpujpeg.c:305-315 loop over the lines on the CPU there).
 * 0 < width_bytes <= both pitches, 0 < rows <= 65536, else UG_HIP_EINVAL before any device call. */
int         ug_hip_memcpy_2d_async(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width_bytes, size_t rows, int kind, ug_hip_stream_t stream);
int         ug_hip_memset_async(void *dst_dev, int value, size_t count, ug_hip_stream_t stream); /* device memory only */
/* Copy lanes: the copy goes onto the ONE upload (download) stream of `device` -- shared by every caller in the process, so that concurrent
 * frames do not split the link between two copies of the same direction -- and is ordered against `stream`: an upload starts after what
 * `then_stream` holds so far and `then_stream` continues after it; a download starts after what `after_stream` holds so far, and
 * ug_hip_stream_sync(after_stream) also waits for it.  The calling thread's current device must be `device`. */
int         ug_hip_upload_ordered(int device, void *dst_dev, const void *src, size_t count, int kind, ug_hip_stream_t then_stream);
int         ug_hip_download_ordered(int device, void *dst_host, const void *src_dev, size_t count, ug_hip_stream_t after_stream);
/* The same with one half of the ordering left to the caller -- what a frame cut into row bands needs (module option bands=<k>: upload of band k+1,
 * kernels of band k and download of band k-1 at the same time, inside ONE frame; the tile fan-out of video_compress.cpp:441-490 is the reference's
 * only intra-frame parallelism):
 *   UG_HIP_COPY_NO_WAIT  upload: the lane does NOT wait for what `then_stream` holds (the caller knows the destination is not in use any more, e.g. the
 *                        next band of a frame whose first band's upload did wait); `then_stream` still continues after the copy
 *   UG_HIP_COPY_NO_JOIN  download: `after_stream` does NOT wait for the copy (its later kernels run beside it); the lane is in order, so a later
 *                        download on the same device WITHOUT this flag makes ug_hip_stream_sync(after_stream) wait for this one too */
#define UG_HIP_COPY_NO_WAIT 1
#define UG_HIP_COPY_NO_JOIN 2
int         ug_hip_upload_ordered_ex(int device, void *dst_dev, const void *src, size_t count, int kind, ug_hip_stream_t then_stream, int flags);
int         ug_hip_download_ordered_ex(int device, void *dst_host, const void *src_dev, size_t count, ug_hip_stream_t after_stream, int flags);
/* the download lane with a 2-D copy (a row band of a picture with a display pitch): geometry as ug_hip_memcpy_2d_async, ordering and flags as above */
int         ug_hip_download_2d_ordered_ex(int device, void *dst_host, size_t dpitch, const void *src_dev, size_t spitch, size_t width_bytes, size_t rows,
                                          ug_hip_stream_t after_stream, int flags);
/* Events (hipEvent_t, timing disabled): record a point of one stream, make another stream wait for it.  What lets a second host thread
 * download the first bands of a picture while the first thread still uploads its last ones (copies from / to pageable host memory block the
 * calling thread: only two threads ever have an upload and a download in flight together). */
typedef void *ug_hip_event_t;
int         ug_hip_event_create(ug_hip_event_t *event);
int         ug_hip_event_destroy(ug_hip_event_t event);                /* NULL: nothing to do */
int         ug_hip_event_record(ug_hip_event_t event, ug_hip_stream_t stream);
int         ug_hip_stream_wait_event(ug_hip_stream_t stream, ug_hip_event_t event);
int         ug_hip_stream_create(ug_hip_stream_t *stream);
int         ug_hip_stream_destroy(ug_hip_stream_t stream);
int         ug_hip_stream_sync(ug_hip_stream_t stream);
const char *ug_hip_last_error_string(void);                            /* cuda_wrapper_last_error_string */
/* NUMA placement of the host threads that feed a GPU (the reference has one worker thread per device, gpujpeg.cpp:446-466, and leaves
 * its placement to the scheduler; on a two-socket, eight-GPU node that is the host-side limit SURVEY.md 8(e) expects).
 * ug_hip_device_numa_node: *node = NUMA node of the device's PCI function (sysfs numa_node), -1 if the platform does not say.
 * ug_hip_bind_thread_to_device: restricts the CALLING THREAD to the CPUs of that node (intersected with its current affinity, never
 * widened); *cpus_bound = CPUs it now runs on, 0 if it was left alone (unknown node, empty intersection).  Call it before the thread
 * allocates its pinned buffers, so that they are first touched on that node.  The *_of_pci / *_to_numa_node forms take the PCI
 * address / node directly and an alternative sysfs root (NULL = "/sys"): what the two above are made of, testable without a GPU. */
int         ug_hip_device_numa_node(int device, int *node);
int         ug_hip_bind_thread_to_device(int device, int *cpus_bound);
int         ug_hip_numa_node_of_pci(const char *bdf, const char *sysfs_root, int *node);
int         ug_hip_bind_thread_to_numa_node(int node, const char *sysfs_root, int *cpus_bound);
/* Average duration in milliseconds of `iters` back-to-back launches of the DXT encoder on
 * `stream`, measured with hipEvents on that stream (cuda_dxt/rgb2dxt1.c:87-108 is the
 * reference's equivalent harness).  Writes ms per launch to *ms_per_launch. */
int         ug_hip_time_dxt_encode(ug_pixfmt_t in, ug_dxt_t out, const void *src, void *dst, int width,
                                   int height, int src_pitch, int frames, size_t src_frame_stride,
                                   size_t dst_frame_stride, int iters, ug_hip_stream_t stream,
                                   float *ms_per_launch);

/* ------------------------------------------------------------------------------------
 * DXT encoders (replace cuda_dxt/cuda_dxt.h:30-89)
 * ---------------------------------------------------------------------------------- */
/* The reference's normative encoders are GLSL shaders (dxt_compress/compress_dxt5ycocg_fp.glsl, compress_dxt1_fp.glsl), and GLSL
 * leaves two things they use to the implementation: the direction of exact .5 ties of round() and the order in which dot(vec3)
 * is summed.  The choice is a run-time option:
 *   UG_DXT_TIES_EVEN  (default) round() ties to even, dot(vec3) summed from the last component: what the shaders compute where
 *                     they can actually be EXECUTED -- Mesa llvmpipe; every block of tests/golden/dxt_glsl_ref.npz (generated by
 *                     running the reference's shader files) is reproduced in this mode.  On the decode side: float -> unorm8
 *                     framebuffer writes of rgba_to_yuv422.glsl / display_dxt1_yuv_fp.glsl tie to even (Mesa again).
 *   UG_DXT_TIES_AWAY  roundf() half away from zero, dot() left to right: the reference's CUDA text (cuda_dxt.cu:106-108,
 *                     122-124), which cannot be executed here; decode side: floor(x * 255 + 0.5).
 * The two differ in ~0.3 % of uniform-random blocks (one endpoint LSB or one palette index). */
#define UG_DXT_TIES_EVEN    0
#define UG_DXT_TIES_AWAY    1
#define UG_DXT_TIES_DEFAULT UG_DXT_TIES_EVEN

/* Fused pixel-format unpack + colour conversion + 4x4 block encode, one pass, no
 * intermediate buffer.  `in` in {RGB, RGBA, UYVY, UYVY_RAW, V210, YUV444}.
 * Requirements: src 16-B aligned, dst 16-B aligned, pitch % 4 == 0 (V210: pitch % 16 == 0 and >= 32 * ceil(width / 12), which
 * vc_get_linesize's 128-byte padding always satisfies; the partial last 12-pixel unit of a line is read whole and encoded in part).
 * ANY width and |height| >= 1 is taken, as dxt_encoder_create does (dxt_compress/dxt_encoder.c:235, called with any tile size by
 * src/video_compress/dxt_glsl.cpp:150-160): the stream holds (width+3)/4 x (height+3)/4 blocks (dxt_get_size, dxt_util.h:59-67);
 * columns past the picture repeat its last column -- the shaders' GL_CLAMP_TO_EDGE fetches (dxt_encoder.c:362-364,
 * compress_dxt5ycocg_fp.glsl:45-55,341), bit for bit -- and lines past the picture repeat its last line (the one deliberate deviation:
 * the reference resamples such a picture vertically and leaves the last block row unrendered, dxt_encoder.c:380 vs :393,653-671;
 * INTEGRATION.md).  With a width that is not a multiple of 4, RGB / YUV444 lines may have any pitch (3 * width bytes), RGBA and
 * UYVY pitch % 4 == 0.  UYVY / V210 need an even width.  These use UG_DXT_TIES_DEFAULT. */
int ug_hip_dxt_encode(ug_pixfmt_t in, ug_dxt_t out, const void *src_dev, void *dst_dev,
                      int width, int height, int src_pitch, ug_hip_stream_t stream);
/* Same, `frames` images per launch (tiles of one frame or consecutive frames):
 * image i at src + i*src_frame_stride -> dst + i*dst_frame_stride. */
int ug_hip_dxt_encode_batch(ug_pixfmt_t in, ug_dxt_t out, const void *src_dev, void *dst_dev,
                            int width, int height, int src_pitch, int frames,
                            size_t src_frame_stride, size_t dst_frame_stride,
                            ug_hip_stream_t stream);
/* Same with the tie rule given explicitly (UG_DXT_TIES_*; anything else: UG_HIP_EINVAL). */
int ug_hip_dxt_encode_batch_ex(ug_pixfmt_t in, ug_dxt_t out, const void *src_dev, void *dst_dev,
                               int width, int height, int src_pitch, int frames,
                               size_t src_frame_stride, size_t dst_frame_stride, int ties,
                               ug_hip_stream_t stream);
/* bytes produced for one image: ((width+3)/4*4) * ((|height|+3)/4*4), half of it for DXT1 (dxt_get_size, dxt_util.h:59-67) */
size_t ug_hip_dxt_size(ug_dxt_t out, int width, int height);

/* Signature-compatible counterparts of cuda_dxt.h (src = tightly packed 3 B/px device
 * buffer; cuda_yuv_* take packed Y,U,V triplets).  These are asynchronous too.  They keep that interface's limits: size_x and
 * |size_y| must be multiples of 4 (cuda_dxt.cu:745 returns -1 otherwise; here UG_HIP_EINVAL). */
int ug_hip_rgb_to_dxt1(const void *src, void *out, int size_x, int size_y, ug_hip_stream_t stream); /* cuda_rgb_to_dxt1, cuda_dxt.h:41 */
int ug_hip_yuv_to_dxt1(const void *src, void *out, int size_x, int size_y, ug_hip_stream_t stream); /* cuda_yuv_to_dxt1, cuda_dxt.h:62 */
int ug_hip_rgb_to_dxt6(const void *src, void *out, int size_x, int size_y, ug_hip_stream_t stream); /* cuda_rgb_to_dxt6, cuda_dxt.h:83 */
int ug_hip_yuv_to_dxt6(const void *src, void *out, int size_x, int size_y, ug_hip_stream_t stream); /* cuda_yuv_to_dxt6, cuda_dxt.h:86 */
int ug_hip_yuv422_to_yuv444(const void *src, void *out, int pix_count, ug_hip_stream_t stream);     /* cuda_yuv422_to_yuv444, cuda_dxt.h:88 */

/* ------------------------------------------------------------------------------------
 * DXT decoders (receiver side; replace dxt_decoder_decompress() behind
 * src/video_decompress/dxt_glsl.c:142-189 and the CPU tool cuda_dxt/dxt62tga.c:24-106)
 * ---------------------------------------------------------------------------------- */
/* `in` in {UG_DXT1, UG_DXT5_YCOCG}; `out` in {UG_PF_RGB, UG_PF_BGR, UG_PF_RGBA, UG_PF_UYVY}.  RGBA output honours
 * rshift/gshift/bshift exactly like the decompress modules' reconfigure() arguments (video_decompress.h:85-100);
 * UYVY follows dxt_compress/rgba_to_yuv422.glsl.  dst_pitch 0 = packed.  Any width, height >= 1 (UYVY: even width): the stream
 * holds (width+3)/4 x (height+3)/4 blocks, width x height pixels are written (dxt_compress/dxt_decoder.c:146-149,368-389). */
int ug_hip_dxt_decode(ug_dxt_t in, ug_pixfmt_t out, const void *src_dev, void *dst_dev, int width, int height,
                      int dst_pitch, int rshift, int gshift, int bshift, ug_hip_stream_t stream);
/* Same with the tie rule given explicitly (UG_DXT_TIES_*): it decides the float -> unorm8 writes of the UYVY output pass and of the
 * DXT1_YUV display matrix; the other outputs do not depend on it. */
int ug_hip_dxt_decode_ex(ug_dxt_t in, ug_pixfmt_t out, const void *src_dev, void *dst_dev, int width, int height,
                         int dst_pitch, int rshift, int gshift, int bshift, int ties, ug_hip_stream_t stream);
/* Device self-test: the decoders divide by the constants 255, 31, 63, 7, 5, 3 with a multiply + two fma (correctly rounded for
 * the numerators a DXT block can produce); this compares every such quotient with the IEEE division. *mismatches must be 0. */
int ug_hip_selftest_dxt_decode(unsigned *mismatches, ug_hip_stream_t stream);
/* Diagnostics (tests, profiling): which DXT5-YCoCg decode path the following ug_hip_dxt_decode calls of this process take --
 * 0 = the product (32-bit fixed point with a guard band; blocks with a value inside the band are decoded again with dxt62tga.c's fp64
 * statements: bit-identical by construction), 1 = the fp64 statements only, 2 = fixed point only (no fallback: shows what the guard is
 * for) -- and, if not NULL, a device counter that every guarded block increments.  Reset with (0, NULL). */
int ug_hip_dxt_decode_debug(int mode, unsigned *flagged_blocks_dev);
/* Same for the encoder: x / 14.0f as multiply + two fma, compared with the IEEE division for x = 0 and every fp32 x in [2^-100, 1] (smaller
 * values take the division itself). */
int ug_hip_selftest_dxt_encode(unsigned *mismatches, ug_hip_stream_t stream);
/* Diagnostics (tests, profiling).  The encoders choose each pixel's ONE decisive comparison of compress_dxt5ycocg_fp.glsl:237-244 /
 * :262-312 from the pixel's position on the palette segment / among the alpha thresholds and evaluate that comparison exactly as the
 * reference does; a wave that holds a block outside the precondition (coincident colour end points over non-flat chroma; luma range < 2^-10) evaluates the
 * reference's full form for all of its blocks.  Returns the number of such waves since the last reset on the current device:
 * [0] colour stage, [1] alpha stage.  Synchronises the device. */
int ug_hip_dxt_encode_stats(unsigned long long full_form_waves[2], int reset);

/* ------------------------------------------------------------------------------------
 * Pixel-format conversion, whole frame on the device (replaces the decoder_t line loop,
 * pixfmt_conv.h:87-88 / pixfmt_conv.c:3041-3125)
 * ---------------------------------------------------------------------------------- */
/* 1 if a kernel exists for in -> out: every pair of decoders[] (== get_decoder_from_to(in, out) != NULL, pixfmt_conv.c:3110-3125) and in == out */
int ug_hip_pixfmt_supported(ug_pixfmt_t in, ug_pixfmt_t out);
/* get_best_decoder_from (pixfmt_conv.c:3126-3172): `candidates` ends with UG_PF_NONE; *out = the candidate the reference's ranking
 * (compare_pixdesc, video_codec.c:1148-1192: keep depth, then subsampling, then colour space; ties to the lower codec_t) puts first among
 * those reachable from `in`.  UG_HIP_EUNSUPP if none is reachable. */
int ug_hip_pixfmt_best(ug_pixfmt_t in, const ug_pixfmt_t *candidates, ug_pixfmt_t *out);
/* rshift/gshift/bshift have decoder_t meaning (honoured for RGBA / RGB outputs, defaults
 * 0/8/16, pixfmt_conv.h:62-65).  Pitches 0 = vc_get_linesize(). */
int ug_hip_pixfmt_convert(ug_pixfmt_t in, ug_pixfmt_t out, const void *src_dev, void *dst_dev,
                          int width, int height, int src_pitch, int dst_pitch,
                          int rshift, int gshift, int bshift, ug_hip_stream_t stream);
/* `frames` images per call, image i at src + i * src_frame_stride -> dst + i * dst_frame_stride.  When the frames follow each other
 * exactly one picture apart on both sides (stride == pitch * height: tiles, frame rings) the batch is ONE launch -- a 4K frame is a
 * 7-13 us launch, which cannot fill 256 CUs on its own; other layouts are converted frame by frame. */
int ug_hip_pixfmt_convert_batch(ug_pixfmt_t in, ug_pixfmt_t out, const void *src_dev, void *dst_dev, int width, int height,
                                int src_pitch, int dst_pitch, int rshift, int gshift, int bshift, int frames,
                                size_t src_frame_stride, size_t dst_frame_stride, ug_hip_stream_t stream);
/* The line converters pixfmt_conv.h:93-101 exports outside decoders[] (their callers reach them by name, e.g. video_capture/screen_x11.c:463,
 * decklink.cpp:1750), whole frame, line by line with the given dst_len (bytes to write per line) and pitches:
 *   "vc_copylineUYVYtoGrayscale" (:927-938), "vc_copylineABGRtoRGB" (:809-843), "vc_copylineBGRAtoRGB" (:845-857),
 *   "vc_copylineToRGBA_inplace" (:907-921; rshift/gshift/bshift are the SOURCE shifts, alpha byte 0; dst may equal src). */
int ug_hip_pixfmt_line_func(const char *func, const void *src_dev, void *dst_dev, int width, int height, int src_pitch, int dst_pitch, int dst_len,
                            int rshift, int gshift, int bshift, ug_hip_stream_t stream);
/* vc_get_linesize (video_codec.c:507-521) for the formats above; UG_HIP_EINVAL for an unknown format or a width outside 1..65536
 * (the reference's int arithmetic wraps there; this never returns a wrapped value) */
int ug_hip_linesize(ug_pixfmt_t fmt, int width);
/* vc_deinterlace (src/video_codec.c:597-664) IN PLACE on a device frame of `lines` lines of `linesize` bytes (pitch == linesize, as there):
 * the linear-blend de-interlace RTDXT applies to INTERLACED_MERGED input before encoding (dxt_glsl.cpp:195-201,291-293), computed as the
 * reference's x86-64 build computes it (its SSE2 bodies: a recursive pavgb blend down the lines; lines < 5: nothing) -- any line size from 16
 * bytes, including those that are no multiple of 16, whose last column reaches into the next line there.  linesize < 16 (where the reference's
 * 16-byte columns overlap themselves and, below 6 bytes, are stored past the frame): UG_HIP_EINVAL.  _batch: `frames` frames frame_stride apart. */
int ug_hip_deinterlace_blend(void *frame_dev, size_t linesize, int lines, ug_hip_stream_t stream);
int ug_hip_deinterlace_blend_batch(void *frame_dev, size_t linesize, int lines, int frames, size_t frame_stride, ug_hip_stream_t stream);

/* packed -> planar (to_planar.h:53-74) */
int ug_hip_uyvy_to_i420(const void *src_dev, int src_pitch, void *y, int y_pitch, void *u, int u_pitch,
                        void *v, int v_pitch, int width, int height, ug_hip_stream_t stream); /* uyvy_to_i420, to_planar.c:343 */
/* v210_to_p010le (to_planar.c:64-155) for every geometry the reference converts: any width, odd last line (:80-83), the
 * width % 6 margin (:85-89 whole last group on interior line pairs -- written past `width`, and with a pitch shorter than
 * roundup6(width) samples the even line's tail stays on the first samples of the odd line, as in the reference; :139-150 the
 * margin of the last one or two lines copied from two lines above).  src_pitch 0 = vc_get_linesize(width, v210); pitches in
 * bytes.  width % 6 != 0 with fewer than 5 lines -> UG_HIP_EUNSUPP (the reference reads in front of its planes there). */
int ug_hip_v210_to_p010le(const void *src_dev, int src_pitch, void *y, int y_pitch, void *uv, int uv_pitch,
                          int width, int height, ug_hip_stream_t stream);

/* Decode-direction shuffles (from_planar.h:58-77 `struct from_planar_data` flattened to scalars; pitch 0 = tightly packed):
 *   ug_hip_yuv420p_to_uyvy      yuv420p_to_uyvy (from_planar.c:583-683) == i420_8_to_uyvy (video_codec.c:1073-1094) for packed planes
 *   ug_hip_yuv422p_to_uyvy      yuv422p_to_uyvy (from_planar.c:391-423)
 *   ug_hip_yuv422p10le_to_v210  yuv422p10le_to_v210 (from_planar.c:296-333): 16-bit little-endian samples holding 10 bits;
 *                               width / 6 groups per line are written, as in the reference
 *   ug_hip_uyvy_to_i422         uyvy_to_i422 (video_codec.c:949-969), planes passed separately */
int ug_hip_yuv420p_to_uyvy(const void *y_dev, int y_pitch, const void *cb_dev, int cb_pitch, const void *cr_dev, int cr_pitch,
                           void *dst_dev, int dst_pitch, int width, int height, ug_hip_stream_t stream);
int ug_hip_yuv422p_to_uyvy(const void *y_dev, int y_pitch, const void *cb_dev, int cb_pitch, const void *cr_dev, int cr_pitch,
                           void *dst_dev, int dst_pitch, int width, int height, ug_hip_stream_t stream);
int ug_hip_yuv422p10le_to_v210(const void *y_dev, int y_pitch, const void *cb_dev, int cb_pitch, const void *cr_dev, int cr_pitch,
                               void *dst_dev, int dst_pitch, int width, int height, ug_hip_stream_t stream);
int ug_hip_uyvy_to_i422(const void *src_dev, int src_pitch, void *y_dev, int y_pitch, void *cb_dev, int cb_pitch, void *cr_dev,
                        int cr_pitch, int width, int height, ug_hip_stream_t stream);
/* uyvy_to_nv12 (to_planar.c:207-302) as the reference's default (-msse4.1) build computes it: chroma of a line pair is
 * (a + b + 1) >> 1 for the first 16 * (width / 16) pixels (_mm_avg_epu8) and (a + b) / 2 for the scalar tail. */
int ug_hip_uyvy_to_nv12(const void *src_dev, int src_pitch, void *y_dev, int y_pitch, void *cbcr_dev, int cbcr_pitch,
                        int width, int height, ug_hip_stream_t stream);

/* The whole of src/from_planar.h and src/to_planar.h by the reference's own function names (these are the
 * building blocks of libavcodec/{from,to}_lavc_vid_conv.c).  The structs repeat the reference's field for field, all pointers
 * are device pointers:
 *   struct ug_from_planar_data == struct from_planar_data (from_planar.h:58-70); `func` = a decode_planar_func_t name (:86-113):
 *     gbrap_to_rgb gbrap_to_rgba gbrp{10,12,16}le_to_{rgb,rgba,rg48,r10k} gbrp{12,16}le_to_r12l rgbpXX_to_rgb
 *     rgbpXXle_to_{rg48,r10k,r12l} yuv444p_to_vuya yuv420p_to_uyvy yuv420_to_i420 yuv422p_to_uyvy yuv422p_to_yuyv
 *     yuv422pXX_to_uyvy yuv422p10le_to_uyvy yuv422p10le_to_v210
 *   struct ug_to_planar_data == struct to_planar_data (to_planar.h:53-59); `func` = a decode_buffer_func_t name (:65-74):
 *     v210_to_p010le y216_to_p010le uyvy_to_nv12 rgba_to_bgra vuya_to_i444 uyvy_to_i420 r12l_to_gbrp12le r12l_to_gbrp16le
 *     r12l_to_rgbp12le            (the packed source is vc_get_linesize(width, codec) per line, as the reference assumes)
 * Same results as the reference byte for byte, including its habits: no masking of bits above the nominal depth,
 * gbrap_to_rgb[a] striding every plane by in_linesize[0], R12L groups written whole.  Differences, both on the safe side:
 * samples past `width` in the last R12L group are taken as 0 (the reference packs uninitialised stack there), and
 * r12l_to_* writes only the samples inside the picture (the reference writes the whole last group of 8).
 * Unknown name or unusable geometry -> UG_HIP_EINVAL. */
struct ug_from_planar_data {
        int width;
        int height;
        void *out_data;
        unsigned out_pitch;
        const void *in_data[4];
        unsigned in_linesize[4];
        int in_depth;      /* for the XX conversions */
        int log2_chroma_h; /* unused here (the reference needs it for its row-band threading only) */
        int rgb_shift[3];  /* *_to_rgba only */
};
struct ug_to_planar_data {
        int width;
        int height;
        void *out_data[4];
        unsigned out_linesize[4];
        const void *in_data;
};
int ug_hip_from_planar(const char *func, const struct ug_from_planar_data *d, ug_hip_stream_t stream);
int ug_hip_to_planar(const char *func, const struct ug_to_planar_data *d, ug_hip_stream_t stream);
int ug_hip_from_planar_supported(const char *func); /* 1 / 0 */
int ug_hip_to_planar_supported(const char *func);

/* The pixel-format converters of src/libavcodec/to_lavc_vid_conv.c and from_lavc_vid_conv.c -- the work the
 * reference reserves a GPU hook for (to_lavc_vid_conv_cuda.h:55-66 `to_lavc_vid_conv_cuda`, from_lavc_vid_conv_cuda.h:55-66
 * `av_to_uv_convert_cuda`; both return NULL there).  Formats are named as the reference names them: UltraGrid codecs by
 * get_codec_name() ("UYVY", "v210", "RGB", "RGBA", "R10k", ...), frame formats by av_get_pix_fmt_name() ("yuv420p", "yuv422p10le",
 * "nv12", "p010le", "gbrp", "xv30le", ...; FFmpeg's enum values are not used, so nothing here needs its headers).
 * struct ug_av_frame carries the AVFrame fields the reference converters read: data, linesize, width, height, colorspace,
 * color_range (numeric values as in libavutil/pixfmt.h: AVCOL_SPC_BT709 1, BT470BG 5, SMPTE170M 6, SMPTE240M 7; AVCOL_RANGE_JPEG 2).
 *   ug_hip_uv_to_av   to_lavc_vid_conv(): the rows of get_uv_to_av_conversion's table (to_lavc_vid_conv.c:1458-1529) -- sources UYVY, v210, RGB,
 *                     RGBA, Y216, Y416, R10k, R12L, RG48;
 *                     `in_data` is vc_get_linesize(width, codec) per line, `out` holds the (device) planes to fill
 *   ug_hip_av_to_uv   av_to_uv_convert(): the rows of av_to_uv_conversions (from_lavc_vid_conv.c:2049-2172) for software frames (not:
 *                     ayuv64le -> UYVY, y210 -> Y216, hardware frames); YCbCr -> RGB picks BT.601 / BT.709 and limited / full range from the
 *                     frame as get_cs_for_conv does (:2614-2658)
 * Results equal the reference functions' byte for byte, slips included (listed in csrc/lavc_conv.hip).  No row: UG_HIP_EUNSUPP.
 * Like the reference functions, some converters work in whole pixel groups (6 for v210, 8 for R12L, 2 for 4:2:2): they may READ up to one
 * group past the end of the last line of a plane (keep the allocation padded, as FFmpeg does) and they WRITE the whole last group of a line
 * when the line size has room for it -- never past the line size. */
struct ug_av_frame {
        void *data[4];
        int linesize[4];
        int width;
        int height;
        int colorspace;
        int color_range;
};
int ug_hip_uv_to_av(const char *uv_codec, const char *av_pixfmt, const void *in_data_dev, const struct ug_av_frame *out, ug_hip_stream_t stream);
int ug_hip_av_to_uv(const char *av_pixfmt, const char *uv_codec, void *dst_dev, int pitch, const struct ug_av_frame *in, const int rgb_shift[3],
                    ug_hip_stream_t stream);
int ug_hip_uv_to_av_supported(const char *uv_codec, const char *av_pixfmt); /* 1 / 0 */
int ug_hip_av_to_uv_supported(const char *av_pixfmt, const char *uv_codec);
/* get_color_coeffs(cs, depth) (color_space.c:149-184): cs 0 = CS_DFL (= BT.709, the reference's default), 1 = BT.601, 2 = BT.709; depth 0 (full range), 8, 10, 12, 16; out = the 14 fields
 * of struct color_coeffs in declaration order */
int ug_hip_color_coeffs(int cs, int depth, int out[14]);
/* compute_color_coeffs(kr, kb, depth) (color_space.c:193-197): the same 14 fields for arbitrary luma weights (host arithmetic in doubles) */
int ug_hip_compute_color_coeffs(double kr, double kb, int depth, int out[14]);

/* ------------------------------------------------------------------------------------
 * JPEG: 8x8 forward DCT + quantisation (the stage libgpujpeg provides behind
 * gpujpeg_encoder_encode, src/video_compress/gpujpeg.cpp:624)
 * ---------------------------------------------------------------------------------- */
/* Annex-K tables scaled by quality (1..100), natural order; comp 0 luma / 1 chroma */
void ug_hip_jpeg_qtable(int quality, int comp, uint8_t table[64]);
/* fp32 reciprocal divisors (AAN scale folded in), natural order */
void ug_hip_jpeg_divisors(const uint8_t qtable[64], float div[64]);
/* One 8-bit plane -> int16 coefficients, zig-zag order, 64 per block, blocks raster order.
 * blocks_w*8 >= width, blocks_h*8 >= height; edge samples replicated.  `div_dev` = 64 floats
 * in device memory.  coef_dev (may be NULL) receives the unquantised fp32 coefficients. */
int ug_hip_jpeg_fdct_quant_plane(const void *plane_dev, int pitch, int width, int height,
                                 int blocks_w, int blocks_h, const float *div_dev,
                                 int16_t *out_dev, float *coef_dev, ug_hip_stream_t stream);
/* Fused UYVY -> 4:2:0 (uyvy_to_i420 rounding) -> FDCT+quantise of Y, Cb, Cr without the
 * I420 round trip through HBM.  Block counts follow 16x16 MCUs: luma (2*mcu_w) x (2*mcu_h)
 * blocks, chroma mcu_w x mcu_h each, mcu_w = ceil(width/16), mcu_h = ceil(height/16).
 * out_y / out_cb / out_cr as above. div_dev = 128 floats (luma then chroma). */
int ug_hip_uyvy_to_jpeg420_coeffs(const void *src_dev, int src_pitch, int width, int height,
                                  const float *div_dev, int16_t *out_y, int16_t *out_cb,
                                  int16_t *out_cr, ug_hip_stream_t stream);
/* Same for 4:2:2, the sampling UltraGrid's GPUJPEG module selects for UYVY input when no `subsampling=` option is
 * given (gpujpeg.cpp:295-302: autoselect = the input codec's subsampling; UYVY is handed over as GPUJPEG_422_U8_P1020,
 * :339): chroma samples are taken as they are (uyvy_to_i422, video_codec.c:949-969), MCU = 16x8:
 * luma (2*mcu_w) x mcu_h blocks, chroma mcu_w x mcu_h each, mcu_w = ceil(width/16), mcu_h = ceil(height/8). */
int ug_hip_uyvy_to_jpeg422_coeffs(const void *src_dev, int src_pitch, int width, int height,
                                  const float *div_dev, int16_t *out_y, int16_t *out_cb,
                                  int16_t *out_cr, ug_hip_stream_t stream);

/* Both of the above over `frames` images per launch (grid.z = image; tiles of a frame or consecutive frames): image i is read at
 * src + i * src_frame_stride and its planes are written i * luma_frame_stride / i * chroma_frame_stride BYTES after out_y /
 * out_cb, out_cr (multiples of 16).  subsampling = 420 or 422.  A 4K frame is a 13 us launch: batching is what fills the GPU. */
int ug_hip_uyvy_to_jpeg42x_coeffs_batch(int subsampling, const void *src_dev, int src_pitch, int width, int height,
                                        const float *div_dev, int16_t *out_y, int16_t *out_cb, int16_t *out_cr, int frames,
                                        size_t src_frame_stride, size_t luma_frame_stride, size_t chroma_frame_stride,
                                        ug_hip_stream_t stream);

/* Complete baseline JPEG encoder (interleaved scan, restart intervals) = FDCT+quantise as above + Huffman coding (T.81
 * Annex K.3 tables) + headers.  Object shape of gpujpeg_encoder_create / _encode / _destroy
 * (src/video_compress/gpujpeg.cpp:353,624,639).  Sampling / input pairs, as the reference module feeds GPUJPEG
 * (gpujpeg.cpp:227-236,295-305,333-344):
 *   420: UG_PF_UYVY (line pairs averaged, uyvy_to_i420) or UG_PF_I420 (planes back to back, passthrough)   JFIF YCbCr
 *   422: UG_PF_UYVY (samples as they are, uyvy_to_i422)                                                     JFIF YCbCr
 *   444: UG_PF_RGB, components stay R, G, B (color_space_internal = GPUJPEG_RGB); written the libjpeg way for JCS_RGB:
 *        Adobe APP14 transform 0, component ids 'R','G','B', quantiser / Huffman table 0 for every component.
 * `encode` is synchronous on `stream` (it returns the stream length).  ug_hip_jpeg_encoder_max_size() is the capacity that can
 * never overflow (every coefficient at its longest code, every byte stuffed: ~10 B per pixel); a smaller out_capacity is allowed:
 * if the stream does not fit, UG_HIP_EINVAL is returned with *out_len = the size it needs and the buffer contents undefined.
 * restart_interval: MCUs per restart interval, 1..65535; 0 = none (one entropy-coded segment, no DRI -- for readers that cannot take restart markers): a lane per
 * block, bit positions by prefix sum, byte stuffing as a second prefix sum; one more host synchronisation per frame; 0.24 ms per 1080p frame, 0.63 ms at 4K against
 * 0.08 / 0.16 ms with restart intervals (profiles/r06_encode_no_restart.txt). */
typedef struct ug_hip_jpeg_encoder ug_hip_jpeg_encoder;
int    ug_hip_jpeg_encoder_create(int width, int height, int quality, int restart_interval, ug_hip_jpeg_encoder **out);
/* subsampling = 420, 422 or 444 (gpujpeg.cpp:406-408 `subsampling=` option); ug_hip_jpeg_encoder_create() is the 420 form. */
int    ug_hip_jpeg_encoder_create_sub(int width, int height, int quality, int restart_interval, int subsampling,
                                      ug_hip_jpeg_encoder **out);
/* The rest of the reference module's encoder options (src/video_compress/gpujpeg.cpp:303-305,396-405):
 *   internal_cs  color_space_internal -- the colour space the samples are CODED in.  UG_JPEG_CS_ASIS (what the two calls above do): the samples as
 *                they come -- R, G, B for RGB input, the BT.709 limited-range Y'CbCr of UltraGrid's UYVY / I420 for the rest.  Otherwise the encoder
 *                converts in front of its forward DCT (ug_hip_jpeg_colour_convert below), taking RGB input as full-range R'G'B' and UYVY as BT.709
 *                limited range: RGB input + a Y'CbCr space -> a JFIF-shaped stream, components 1, 2, 3, chroma tables for Cb and Cr (Y601full IS
 *                JFIF); UYVY input + BT.601 (either range) -> the usual 4:2:x stream of the converted samples; UG_JPEG_CS_RGB with RGB input and
 *                UG_JPEG_CS_YCBCR_BT709 with UYVY / I420 input = UG_JPEG_CS_ASIS.  Not offered (UG_HIP_EUNSUPP): R, G, B components with
 *                subsampling 420 / 422, I420 with a conversion (ug_hip_yuv420p_to_uyvy first, as the module does).  The colour stage is UNPINNED towards libgpujpeg like the FDCT: published BT.601 /
 *                BT.709 definitions, fp32.
 *   flags        UG_JPEG_NONINTERLEAVED (subsampling 444 only): one scan per component (T.81 A.2.2; restart intervals count blocks of the scan's
 *                component) -- the reference's DEFAULT for RGB input (interleaved = 0 unless `:interleaved`, gpujpeg.cpp:303); the header then
 *                carries what those scans use (RGB: quantiser and Huffman table 0 only).  Three coder launches, each going on where the one before
 *                ended (one synchronisation, no intermediate buffer); the single interleaved scan is ONE fused kernel and faster.
 *                UG_JPEG_INPUT_UYVY (subsampling 444 only): the 4:4:4 encoder is fed UYVY instead of RGB (`-c jpeg:subsampling=444` on a 4:2:2
 *                source, gpujpeg.cpp:297-302 with GPUJPEG_422_U8_P1020 input): every pixel takes its pair's chroma, and the samples are coded as
 *                they are (UG_JPEG_CS_ASIS / _BT709: a Y'CbCr stream like the one RGB input + _BT709 gives), converted to BT.601 (either range),
 *                or converted to R, G, B (UG_JPEG_CS_RGB: the R,G,B stream of RGB input, from a 4:2:2 source). */
#define UG_JPEG_CS_ASIS                0
#define UG_JPEG_CS_RGB                 1 /* GPUJPEG_RGB: full-range R'G'B' */
#define UG_JPEG_CS_YCBCR_BT601         2 /* GPUJPEG_YCBCR_BT601: limited range (16-235 / 16-240) */
#define UG_JPEG_CS_YCBCR_BT601_256LVLS 3 /* GPUJPEG_YCBCR_BT601_256LVLS: full range -- the JFIF colour space */
#define UG_JPEG_CS_YCBCR_BT709         4 /* GPUJPEG_YCBCR_BT709: limited range */
#define UG_JPEG_NONINTERLEAVED         1
#define UG_JPEG_INPUT_UYVY             2
int    ug_hip_jpeg_encoder_create_ex(int width, int height, int quality, int restart_interval, int subsampling, int internal_cs, int flags,
                                     ug_hip_jpeg_encoder **out);
/* The colour stage by itself.  m[12]: out_i = m[4 i] * in_0 + m[4 i + 1] * in_1 + m[4 i + 2] * in_2 + m[4 i + 3] on 8-bit code values, cs_in -> cs_out
 * (UG_JPEG_CS_RGB .. UG_JPEG_CS_YCBCR_BT709), derived in double from Kr / Kb of BT.601 (0.299, 0.114) and BT.709 (0.2126, 0.0722) and the 219 / 224
 * (limited) or 255 / 255 (256 levels) code ranges.  convert: fmt = UG_PF_RGB (3 bytes per pixel, whatever the three components mean) or UG_PF_UYVY
 * (every pixel with its pair's chroma, the two chroma results of a pair averaged); fp32, one multiply-add at a time, clamp, round to nearest even;
 * pitch 0 = packed. */
int    ug_hip_jpeg_colour_matrix(int cs_in, int cs_out, float m[12]);
int    ug_hip_jpeg_colour_convert(ug_pixfmt_t fmt, int cs_in, int cs_out, const void *src_dev, int src_pitch, void *dst_dev, int dst_pitch, int width, int height,
                                  ug_hip_stream_t stream);
void   ug_hip_jpeg_encoder_destroy(ug_hip_jpeg_encoder *enc);
size_t ug_hip_jpeg_encoder_max_size(const ug_hip_jpeg_encoder *enc);
int    ug_hip_jpeg_encoder_encode(ug_hip_jpeg_encoder *enc, ug_pixfmt_t in, const void *src_dev, int src_pitch,
                                  void *out_dev, size_t out_capacity, size_t *out_len, ug_hip_stream_t stream);
/* `frames` (1..16) frames of the same geometry in ONE call: one launch sequence over all of them (UYVY / RGB input: ONE kernel does the
 * forward DCT, the quantiser, the Huffman coding and the byte stuffing of every frame, a second small one assembles the streams), one
 * synchronisation, `frames` lengths.  Frame f is read at src_dev + f * src_stride and its stream
 * written at out_dev + f * out_stride (a multiple of 16, >= out_capacity = what one stream may take); out_len[f] = its length; with frames > 1
 * a stream that does not fit is reported per frame -- out_len[f] > out_capacity = the size it needs, the other streams are complete and
 * the call succeeds (the one-frame call returns UG_HIP_EINVAL for it).  Every
 * stream is byte-identical to what ug_hip_jpeg_encoder_encode writes for that frame.  (gpujpeg.cpp:617-631 encodes one frame per
 * call; this is for callers that hold several queued frames: 14 us per 4K 4:2:0 frame at 8 per call against 38 us one by one.) */
int    ug_hip_jpeg_encoder_encode_batch(ug_hip_jpeg_encoder *enc, ug_pixfmt_t in, int frames, const void *src_dev, int src_pitch,
                                        size_t src_stride, void *out_dev, size_t out_stride, size_t out_capacity, size_t *out_len,
                                        ug_hip_stream_t stream);

/* ------------------------------------------------------------------------------------
 * JPEG decoder (receive side: gpujpeg_decoder_create / _decode / _destroy behind
 * src/video_decompress/gpujpeg.c:74-140,292-301)
 * ---------------------------------------------------------------------------------- */
/* Baseline JPEG (8-bit, Huffman; 3 components at 4:4:4 / 4:2:2 / 4:2:0, interleaved or one scan per component; restart intervals make the
 * entropy-coded data parallel: one lane per restart segment) -> `out` in device memory:
 *   YCbCr streams -> UG_PF_UYVY (4:2:2: samples as they are; 4:2:0: chroma lines repeated, yuv420p_to_uyvy; 4:4:4: chroma pairs averaged),
 *                    UG_PF_RGB / UG_PF_RGBA (the UYVY form through vc_copylineUYVYtoRGB[A]: BT.709 limited range, as UltraGrid codes it),
 *                    UG_PF_I420 (4:2:0 streams: planes back to back);
 *   R,G,B streams (Adobe APP14 transform 0 or component ids 'R','G','B': what `-c jpeg` writes for RGB input) -> UG_PF_RGB / UG_PF_RGBA
 *                    directly, UG_PF_UYVY through vc_copylineRGBtoUYVY's arithmetic.
 *   one-component (greyscale) streams: a Y'CbCr picture with both chroma planes at 128, every output above (what another sender's grey MJPEG needs).
 * The component planes equal libjpeg's bit for bit (integer IDCT jidctint).  `jpeg_host` is host memory (compressed frames arrive from the
 * network); everything after the header parse is asynchronous on `stream` (a stream in pinned memory is read by the copy engine when the stream
 * gets there: keep it until then; pageable memory is staged before the call returns; a scan WITHOUT restart intervals -- one segment, decoded by a lane per 1024
 * bits of it with the decoder's states handed from piece to piece until they settle -- synchronises with the host between its launches).  UG_PF_NONE: decode to the internal planes only
 * (ug_hip_jpeg_decoder_plane).  Not a baseline stream / unsupported layout (incl. a second frame header, or a table redefined between
 * the scans of a one-scan-per-component stream): UG_HIP_EUNSUPP.  Damage inside the entropy-coded data is not
 * an error: a segment ends at its first marker, a missing one decodes as an empty one (what a sequential decoder does).  A decoder object
 * holds the work buffers of one frame in flight: one object per thread / per frame in flight. */
typedef struct ug_hip_jpeg_decoder ug_hip_jpeg_decoder;
int  ug_hip_jpeg_decoder_create(ug_hip_jpeg_decoder **out);
void ug_hip_jpeg_decoder_destroy(ug_hip_jpeg_decoder *dec);
/* header only: subsampling = 444 / 422 / 420 / 400 (greyscale); any pointer may be NULL */
int  ug_hip_jpeg_read_info(const void *jpeg_host, size_t len, int *width, int *height, int *subsampling, int *is_rgb, int *restart_interval);
int  ug_hip_jpeg_decoder_decode(ug_hip_jpeg_decoder *dec, const void *jpeg_host, size_t len, ug_pixfmt_t out, void *dst_dev, int dst_pitch,
                                int rshift, int gshift, int bshift, ug_hip_stream_t stream);
/* The same with the picture size the caller sized `dst_dev` for (0 = not checked): a stream whose frame header says otherwise is refused
 * with UG_HIP_EINVAL before anything is written -- streams come from the network, and the size check must not depend on two separate
 * header parses (ug_hip_jpeg_read_info, then this) agreeing. */
int  ug_hip_jpeg_decoder_decode_sized(ug_hip_jpeg_decoder *dec, const void *jpeg_host, size_t len, int expect_width, int expect_height,
                                      ug_pixfmt_t out, void *dst_dev, int dst_pitch, int rshift, int gshift, int bshift, ug_hip_stream_t stream);
/* component plane of the last decode (device memory, padded to whole MCUs; width / height = the component's own size) */
int  ug_hip_jpeg_decoder_plane(const ug_hip_jpeg_decoder *dec, int component, const void **plane_dev, int *pitch, int *width, int *height);

#ifdef __cplusplus
}
#endif
#endif /* UG_MI355X_H */
