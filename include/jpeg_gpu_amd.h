/* jpeg_gpu_amd — MI355X-native JPEG block-decode path: C-ABI boundary.
 *
 * This header is the drop-in boundary.  It has three parts:
 *
 *  1. The data model the reference harness already speaks (struct image,
 *     image_plane, jpeg_header, jpeg_component, jpeg_quant, jpeg_info, the
 *     jpeg_decode_out stage enum and the jpeg_decode_ctx_vtbl plugin table).
 *     These are restated field-for-field because they ARE the interface:
 *       image / image_plane ......... reference src/image.h:25-51
 *       jpeg_quant .. jpeg_info ..... reference src/jpeg_info.h:37-71
 *       jpeg_decode_out, vtbl ....... reference src/jpeg_wrap.h:22-51
 *     If the reference's own headers were included first (their include
 *     guards _image_H / _jpeg_info_H / _jpeg_wrap_H are defined) the
 *     restatement is skipped, so a maintainer can include both.
 *
 *  2. HIPJPEG_DECODE_CTX_VTBL — the plugin instance that sits where
 *     XJPEG_DECODE_CTX_VTBL / LIBJPEG_DECODE_CTX_VTBL sit
 *     (reference src/jpeg_wrap.h:53-54, selected at src/jpeg_gpu.c:545-557).
 *
 *  3. jga_* entry points: the pieces of the hot path as plain C calls (layout,
 *     host entropy decode, device launches on resident coefficient planes,
 *     the pipelined batch decoder).  No torch / HIP types appear: device
 *     pointers are void*, streams are void* (a hipStream_t).
 *
 * All functions returning int use the reference's convention
 * (src/jpeg_wrap.c:269-283, 329-339): 0 = EXIT_SUCCESS, 1 = EXIT_FAILURE, with
 * a one-line message on stderr (also retrievable via jga_last_error()).
 */
#ifndef JPEG_GPU_AMD_H
#define JPEG_GPU_AMD_H (1)

#include <stddef.h>
#include <stdint.h>
#include <string.h>   /* (memset in jga_pipeline_config_init) */

#ifdef __cplusplus
extern "C" {
#endif
/* everything declared here is exported from libjpeg_gpu_amd.so (which is built
 * with -fvisibility=hidden) */
#pragma GCC visibility push(default)

/* ------------------------------------------------------------------------ */
/* 1. Data model (ABI-identical to the reference; sizes asserted in csrc/layout.c)   */
/* ------------------------------------------------------------------------ */

#if !defined(_jpeg_info_H)
# define NCOMPS_MAX (3)
# define NQUANT_MAX (4)

typedef enum jpeg_subsamp {
  JPEG_SUBSAMP_UNKNOWN,
  JPEG_SUBSAMP_444,
  JPEG_SUBSAMP_422,
  JPEG_SUBSAMP_420,
  JPEG_SUBSAMP_440,
  JPEG_SUBSAMP_411,
  JPEG_SUBSAMP_MONO,
  JPEG_SUBSAMP_MAX
} jpeg_subsamp;

typedef struct jpeg_quant {
  int valid;
  unsigned char bits;          /* 8 or 16 */
  unsigned short tbl[64];      /* NATURAL (de-zigzagged) order, xjpeg.c:240,245 */
} jpeg_quant;

typedef struct jpeg_component {
  int hblocks;                 /* nhmb*hsamp: MCU-padded width in blocks  */
  int vblocks;                 /* nvmb*vsamp: MCU-padded height in blocks */
  int hsamp;
  int vsamp;
  jpeg_quant *quant;           /* points INTO the owning jpeg_header */
} jpeg_component;

typedef struct jpeg_header {
  int bits;
  int width;
  int height;
  int ncomps;
  jpeg_subsamp subsamp;
  int restart_interval;
  jpeg_component comp[NCOMPS_MAX];
  jpeg_quant quant[NQUANT_MAX];
} jpeg_header;

typedef struct jpeg_info {
  int size;
  unsigned char *buf;          /* caller-owned JPEG file bytes */
} jpeg_info;
#endif /* _jpeg_info_H */

#if !defined(_image_H)
# define NPLANES_MAX (3)

typedef struct image_plane {
  int bitdepth;
  unsigned char xdec;
  unsigned char ydec;
  int xstride;
  int ystride;
  unsigned short width;        /* MCU-padded, = hblocks*8 */
  unsigned short height;       /* MCU-padded, = vblocks*8 */
  unsigned char *data;         /* ystride*height u8 samples */
  short *coef;                 /* this plane's slice of image.coef */
  int cstride;                 /* rows of luma-width block rows it occupies */
  int packed;
  int *index;
} image_plane;

typedef struct image {
  unsigned short width;        /* true size */
  unsigned short height;
  int nplanes;
  image_plane plane[NPLANES_MAX];
  short *coef;                 /* packed coefficient planes ("quant"/"dct") */
  int packed;
  int *index;
  unsigned char *pixels;       /* width*height*3 interleaved RGB (1 B/px grey) */
} image;
#endif /* _image_H */

#if !defined(_jpeg_wrap_H)
typedef struct jpeg_decode_ctx jpeg_decode_ctx;   /* opaque */

typedef enum jpeg_decode_out {
  JPEG_DECODE_PACK,
  JPEG_DECODE_QUANT,
  JPEG_DECODE_DCT,
  JPEG_DECODE_YUV,
  JPEG_DECODE_RGB,
  JPEG_DECODE_OUT_MAX
} jpeg_decode_out;

typedef jpeg_decode_ctx *(*jpeg_decode_alloc_func)(jpeg_info *info);
typedef int (*jpeg_decode_header_func)(jpeg_decode_ctx *dec,
 jpeg_header *header);
typedef int (*jpeg_decode_image_func)(jpeg_decode_ctx *dec, image *img,
 jpeg_decode_out out);
typedef void (*jpeg_decode_reset_func)(jpeg_decode_ctx *dec, jpeg_info *info);
typedef void (*jpeg_decode_free_func)(jpeg_decode_ctx *dec);

typedef struct jpeg_decode_ctx_vtbl {
  jpeg_decode_alloc_func decode_alloc;
  jpeg_decode_header_func decode_header;
  jpeg_decode_image_func decode_image;
  jpeg_decode_reset_func decode_reset;
  jpeg_decode_free_func decode_free;
} jpeg_decode_ctx_vtbl;
#endif /* _jpeg_wrap_H */

/* ------------------------------------------------------------------------ */
/* 2. The plugin instance                                                    */
/* ------------------------------------------------------------------------ */

/* Replaces XJPEG_DECODE_CTX_VTBL (reference src/jpeg_wrap.c:352-358).
 * Call order is the reference's (src/jpeg_gpu.c:612-613, 1215, 1231-1237):
 * alloc -> header -> [image_init] -> image, then per frame reset -> header ->
 * image, free at exit.  Stages: QUANT and DCT fill img->coef on the host (same
 * bytes as the reference, src/xjpeg.c:550-563); YUV fills every
 * img->plane[i].data and RGB fills img->pixels, both computed on the GPU.
 * PACK fills img->coef with the RLE words, img->index with the block starts and
 * sets plane[i].packed / img->packed (src/xjpeg.c:484-496, 513-519, 531-535). */
extern const jpeg_decode_ctx_vtbl HIPJPEG_DECODE_CTX_VTBL;

/* The comparison backend: the platform's libjpeg behind the same table, as the
 * reference's LIBJPEG_DECODE_CTX_VTBL (src/jpeg_wrap.c:56-252, jpeg_wrap.h:53).
 * CPU only, stages QUANT / YUV / RGB (jpeg_wrap.c:134-228).  libjpeg.so.8 is bound
 * at run time; decode_alloc returns NULL when it is not installed
 * (jga_libjpeg_available() == 0).  Not on the hot path: bench.py times it beside the
 * MI355X path and the harness offers it as `-i libjpeg`. */
/* Exported under a name of its own: a maintainer who adds this library to the reference's build
 * keeps linking jpeg_wrap.o, which DEFINES the data symbol LIBJPEG_DECODE_CTX_VTBL
 * (src/jpeg_wrap.c:246-252) — two definitions of one symbol would be interposed silently by a
 * shared library and refused by a static one.  Code that does not include the reference's
 * jpeg_wrap.h may still use the reference's name: it is an alias then. */
extern const jpeg_decode_ctx_vtbl JGA_LIBJPEG_DECODE_CTX_VTBL;
#if !defined(_jpeg_wrap_H)
# define LIBJPEG_DECODE_CTX_VTBL JGA_LIBJPEG_DECODE_CTX_VTBL
#endif
int jga_libjpeg_available(void);

/* ------------------------------------------------------------------------ */
/* 3. jga_* C entry points                                                   */
/* ------------------------------------------------------------------------ */

const char *jga_version(void);
/* Last error message of the calling thread ("" if none). */
const char *jga_last_error(void);

/* --- layout: replaces image_init / image_zero / image_clear
 *     (reference src/image.c:24-97, 99-112, 114-124).  Same fields, same
 *     sizes, 16-byte aligned buffers; buffers come from the library's own
 *     allocator so only jga_image_clear may free them. ------------------- */
int  jga_image_init(image *img, jpeg_header *header);
void jga_image_zero(image *img);
void jga_image_clear(image *img);

/* Geometry of the packed coefficient buffer and of the outputs for one frame,
 * derived from a jpeg_header exactly as image_init + xjpeg.c:556-561 imply
 * (SURVEY.md Appendix B).  All offsets in shorts. */
typedef struct jga_plane_geom {
  int hblocks;                 /* plane width  in blocks (MCU padded) */
  int vblocks;                 /* plane height in blocks (MCU padded) */
  int xdec;
  int ydec;
  int cstride;                 /* packed rows of RS shorts */
  int qidx;                    /* which of qtab[0..2] this plane uses (= plane#) */
  long long coef_off;          /* offset of plane base in the coef buffer */
  long long data_off;          /* offset of plane in a concatenated YUV buffer */
} jga_plane_geom;

typedef struct jga_geom {
  int width;                   /* true size */
  int height;
  int nplanes;                 /* 1 or 3 */
  int subsamp;                 /* jpeg_subsamp */
  int w0;                      /* luma padded width in pixels; RS = w0*8 */
  int nhmb;                    /* MCUs per row */
  int nvmb;                    /* MCU rows */
  int restart_interval;
  long long coef_shorts;       /* size of the packed coefficient buffer */
  long long coef_blocks;       /* real coded blocks (sum hblocks*vblocks) */
  long long yuv_bytes;         /* sum of padded plane sizes */
  long long rgb_bytes;         /* width*height*nplanes (3 B/px colour, 1 grey) */
  jga_plane_geom plane[NPLANES_MAX];
} jga_geom;

int jga_geom_from_header(jga_geom *g, const jpeg_header *header);
/* Offset (in shorts) of block (bx,by) of plane p: xjpeg.c:556-561. */
long long jga_block_offset(const jga_geom *g, int p, int bx, int by);

/* --- host entropy stage: replaces xjpeg_init/xjpeg_decode_header/
 *     xjpeg_decode_image(QUANT|DCT) (reference src/xjpeg.c:449-632, 704-780).
 *     Bounds-checked, restart-aware; writes de-zigzagged int16 blocks in the
 *     Appendix-B layout.  `dequant` = 0 gives the QUANT stage (raw levels),
 *     1 gives the DCT stage (level*q truncated to int16). ---------------- */
int jga_parse_header(const unsigned char *buf, int size, jpeg_header *header);
int jga_entropy_decode(const unsigned char *buf, int size,
 const jga_geom *g, short *coef, int dequant);
/* PACK wire format (xjpeg.c:484-496, 513-519, 531-535): RLE words + per-block
 * start index (index laid out per plane as image_init does, src/image.c:93-94).
 * Returns the number of words via *nwords and, if plane_words != NULL, the
 * words each plane contributed (the reference's image_plane.packed). */
int jga_entropy_decode_pack(const unsigned char *buf, int size,
 const jga_geom *g, short *pack, long long pack_cap, int *index,
 long long *nwords, long long *plane_words);

/* --- one frame in bands of MCU rows that decode independently (restart intervals:
 *     reference src/xjpeg.c:593-629, DC predictors reset 612-618, DRI 412-420).  A frame whose
 *     restart intervals tile its MCU rows (interval = a row of MCUs, or any length of which a whole
 *     number fills some number of rows) is cut where intervals begin on a row boundary;
 *     jga_band_file() writes one band as a baseline JPEG file of its own (the frame's marker
 *     segments, SOF0 height = the band's rows, its entropy-coded bytes with the RSTn counters
 *     renumbered from 0, EOI), which every decode entry point here — and the reference's
 *     xjpeg_decode_image() — takes like any other file.  Its pixels are rows [y0, y0 + rows) of
 *     the frame's, bit for bit.  One process per GPU decodes band `rank` of `world`; no collective.
 *     A frame without restart markers is one band.  Host code only. ---- */
typedef struct jga_band {
  int index, count;            /* band `index` of `count` */
  int mcu_row0, mcu_rows;      /* the MCU rows of the frame it covers */
  int y0, rows;                /* = pixel rows [y0, y0 + rows) of the frame */
  int first_interval;          /* number of its first restart interval in the frame */
  int reserved_;
  long scan_off, scan_bytes;   /* its entropy-coded bytes in the frame's file (no marker before or after) */
} jga_band;
/* bands[0 .. count): returns how many were made (fewer than `count` when the frame has fewer
 * rows of MCUs that can stand alone), or < 0 (jga_last_error()). */
int jga_band_plan(const unsigned char *file, long size, int count, jga_band *bands);
/* The band as a file: returns its length (out == NULL: the length `out` must hold), or < 0. */
long jga_band_file(const unsigned char *file, long size, const jga_band *band,
 unsigned char *out, long cap);

/* --- device stage: dequantise + row IDCT + column IDCT + level shift/clamp
 *     (+ chroma upsample + YCbCr->RGB) on coefficient planes RESIDENT IN HBM.
 *     Replaces the three GLSL passes res/horz_quant_yuv.fs.glsl:81-99,
 *     res/vert.fs.glsl:79-101, res/unyuv.fs.glsl:17-50 (and ungrey) and their
 *     driver src/jpeg_gpu.c:1312-1365; arithmetic is the CPU path's
 *     (src/dct.c:100-121, src/xjpeg.c:501-503,524-527,565-584).
 *
 *     d_coef : nimages coefficient buffers, image i at d_coef + i*coef_stride
 *              (shorts), each in the Appendix-B layout of `g`.
 *     d_qtab : nimages * 3 * 64 uint16, natural order, table of plane p of
 *              image i at d_qtab[(i*3+p)*64].  (`dequant_on_device`=0 means
 *              the coefficients are already dequantised = DCT stage input and
 *              d_qtab may be NULL.)
 *     d_rgb  : image i at d_rgb + i*rgb_stride, row pitch width*nplanes.
 *     d_yuv  : image i at d_yuv + i*yuv_stride, planes concatenated at
 *              g->plane[p].data_off, each padded, ystride = plane width.
 *     stream : hipStream_t (NULL = default stream).  Asynchronous.
 *
 *     Sampling: any factors of 1, 2, 4 with luma the finest plane (the reference's pass 3
 *     reads Y undecimated, res/unyuv.fs.glsl:23-28; src/xjpeg.c:384-391).  Cb and Cr
 *     decimated alike — every usual file — run one fused kernel; Cb and Cr decimated
 *     DIFFERENTLY (res/unyuv.fs.glsl:6-9 allows it) make jga_idct_rgb_batch run the YUV
 *     stage and pass 3 through a scratch buffer and return with the pixels complete. ---- */
int jga_device_count(void);
int jga_idct_rgb_batch(const jga_geom *g, int nimages,
 const short *d_coef, long long coef_stride, const unsigned short *d_qtab,
 int dequant_on_device, unsigned char *d_rgb, long long rgb_stride,
 void *stream);
int jga_idct_yuv_batch(const jga_geom *g, int nimages,
 const short *d_coef, long long coef_stride, const unsigned short *d_qtab,
 int dequant_on_device, unsigned char *d_yuv, long long yuv_stride,
 void *stream);
/* Pass 3 alone: YUV-stage planes resident in HBM (layout of d_yuv above) ->
 * RGB (layout of d_rgb above).  The reference's res/yuv.fs.glsl:16-24 /
 * res/unyuv.fs.glsl:12-16, 29-31, 39-41, 48 on u8 planes, for callers that stop
 * the decode at JPEG_DECODE_YUV.  Same results as jga_idct_rgb_batch. */
int jga_yuv_rgb_batch(const jga_geom *g, int nimages,
 const unsigned char *d_yuv, long long yuv_stride, unsigned char *d_rgb,
 long long rgb_stride, void *stream);
/* PACK wire format expanded on the device: the words and
 * per-block start indices produced by jga_entropy_decode_pack() (reference
 * producer src/xjpeg.c:484-496, 513-519, 531-535), resident in HBM, become the
 * QUANT-stage planes the two launches above read — the job of the reference's
 * first PACK pass, res/horz_pack_yuv.fs.glsl:94-127 (12-bit sign extension,
 * `j += run + 1`, de-zigzag).  Image i: words at d_pack + i*pack_stride (of which
 * pack_words may be read), indices at d_index + i*index_stride
 * (jga_index_count(g) ints, planes back to back, raster per plane as
 * src/image.c:93-94), planes at d_coef + i*coef_stride.  Only real blocks are
 * written.  Asynchronous on `stream`. */
long long jga_index_count(const jga_geom *g);
int jga_unpack_batch(const jga_geom *g, int nimages,
 const unsigned short *d_pack, long long pack_stride, long long pack_words,
 const int *d_index, long long index_stride, short *d_coef, long long coef_stride,
 void *stream);
/* Name of the kernel jga_idct_rgb_batch / jga_idct_yuv_batch launch for `g` (for profiles). */
const char *jga_kernel_name(const jga_geom *g, int rgb);

/* Thin device-memory helpers so C/ctypes callers need no HIP binding. */
void *jga_device_malloc(size_t bytes);
void  jga_device_free(void *p);
void *jga_host_malloc_pinned(size_t bytes);
void  jga_host_free_pinned(void *p);
/* Pin memory the caller already owns (hipHostRegister): ingest buffers a pipeline may DMA from
 * directly (jga_job.pinned).  Returns EXIT_SUCCESS / EXIT_FAILURE. */
int   jga_host_register(void *p, size_t bytes);
int   jga_host_unregister(void *p);
int   jga_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream);
int   jga_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream);
/* The same two copies through a pinned bounce buffer of the library's, synchronously (NULL stream): for host buffers that
 * are short-lived ordinary memory — a copy that NAMES such memory leaves the runtime's pinning of it cached (read-only
 * for a source), and the range may come back from the allocator as somebody's pixel buffer (csrc/device_api.cpp). */
int   jga_upload_staged(void *d_dst, const void *h_src, size_t bytes);
int   jga_download_staged(void *h_dst, const void *d_src, size_t bytes);
int   jga_device_memset(void *dst, int value, size_t bytes, void *stream);
int   jga_stream_sync(void *stream);
int   jga_set_device(int dev);
/* "dddd:bb:dd.f" of device `dev` (len >= 13): a launcher that runs one process per GPU
 * finds the GPU's NUMA node and local CPUs under /sys/bus/pci/devices/<id>/ with it. */
int   jga_device_pci_bus_id(int dev, char *buf, int len);
void *jga_stream_create(void);
void  jga_stream_destroy(void *stream);
/* jga_idct_rgb_batch / jga_idct_yuv_batch with coefficient 0 of every block taken from d_dc
 * (jga_huff_decode_split's array; NULL = from the planes, i.e. the plain calls). */
int jga_idct_rgb_batch_dc(const jga_geom *g, int nimages, const short *d_coef,
 long long coef_stride, const short *d_dc, long long dc_stride, const unsigned short *d_qtab,
 int dequant_on_device, unsigned char *d_rgb, long long rgb_stride, void *stream);
int jga_idct_yuv_batch_dc(const jga_geom *g, int nimages, const short *d_coef,
 long long coef_stride, const short *d_dc, long long dc_stride, const unsigned short *d_qtab,
 int dequant_on_device, unsigned char *d_yuv, long long yuv_stride, void *stream);
/* Time `reps` back-to-back launches of the rgb (or yuv) batch kernel with HIP
 * events on `stream`; returns average milliseconds per launch in *ms. */
int jga_time_idct_batch(const jga_geom *g, int nimages, const short *d_coef,
 long long coef_stride, const unsigned short *d_qtab, int dequant_on_device,
 unsigned char *d_out, long long out_stride, int rgb, int reps, void *stream,
 float *ms);

/* `reps` device-to-device copies (hipMemcpyDtoDAsync) of `bytes`, HIP events on `stream` around them: the copy
 * ceiling the bench prints beside the kernels' rates; average milliseconds per copy in *ms. */
int jga_time_device_copy(void *d_dst, const void *d_src, size_t bytes, int reps, void *stream, float *ms);
/* The same volume moved by a kernel of this library that does nothing else (csrc/copy_kernel.hip: `grid` workgroups
 * of 256 lanes, 16 bytes per lane per trip, non-temporal stores); bytes a multiple of 16. */
int jga_time_kernel_copy(void *d_dst, const void *d_src, size_t bytes, int grid, int reps, void *stream, float *ms);

/* --- pipelined batch decoder (build addition; SURVEY.md §8b "batch/async
 *     entry"): N host entropy threads -> pinned ring -> H2D on a copy stream
 *     -> fused kernel on a compute stream (-> optional D2H).  One pipeline
 *     per GPU; images are independent, no collectives. ------------------- */
typedef struct jga_pipeline jga_pipeline;

typedef struct jga_pipeline_config {
  int struct_size;             /* sizeof(jga_pipeline_config) and sizeof(jga_job) of the header the CALLER was */
  int job_size;                /* built against (jga_pipeline_config_init fills them in): jga_pipeline_create
                                * turns a caller from another revision away instead of walking its job
                                * array with the wrong stride */
  int device;                  /* HIP device ordinal */
  int nthreads;                /* host entropy threads (0 = hardware default) */
  int depth;                   /* transport 2: lanes (groups in flight), 0 = 6.  Transports 0/1
                                * always run two pinned slots per thread; depth is ignored */
  int out;                     /* JPEG_DECODE_YUV or JPEG_DECODE_RGB */
  int copy_back;               /* 1: D2H into caller's host buffers */
  long long max_coef_shorts;   /* slot capacity (0 = sized on first submit) */
  long long max_out_bytes;
  int transport;               /* what crosses PCIe: 0 = dense QUANT planes,
                                * 1 = PACK words + block index, expanded by jga_unpack_batch,
                                * 2 = the entropy-coded bytes: no host Huffman, jga_huff_* on
                                *     the GPU (depth = lanes in flight, default 6) */
  int batch;                   /* transport 2: 4K frames per GPU entropy batch (0 = 48); smaller
                                * frames fill a group to about the same pixel count (up to 16x
                                * as many); jobs are grouped by geometry in arrival order */
  int unstuff;                 /* transport 2, where stuffing and RSTn markers are removed: 0 = auto (on
                                * the GPU when the host side has 8 cores or fewer to count on — the
                                * smaller of the process's CPU grant and `nthreads` — and then groups
                                * whose jobs are all `pinned` are uploaded by DMA straight from the
                                * callers' buffers; else on the host: one core unstuffs ~12 GB/s, as
                                * fast as it could copy), 1 = host, 2 = GPU (jga_huff_set_device_unstuff) */
  int spin_waits;              /* 0: lanes waiting for the device poll and sleep (csrc/host_wait.h), 1: they
                                * spin in the runtime's waits (a core each; the lowest latency) */
  int trace;                   /* 1: a timeline of every run on stderr */
  /* (How transport 2 schedules its lanes — groups per lane, turns on the link and on the device, the ramp of a long
   *  job's first groups — is not configuration: every other value of those fields measured slower or the same in
   *  rounds 3-4, profiles/r4_host_side_steps.md.  Builds made with -DJGA_TUNING still read them from JGA_PIPE_*
   *  environment variables, csrc/jga_tune.h.) */
  /* --- ingest of caller-owned PAGEABLE buffers without a host copy, where the scan clean-up runs on the device
   *     (unstuff 2, 0 with few cores; a host that cleans up reads every byte anyway).  The pipeline registers the
   *     callers' JPEG buffers with the device (hipHostRegister: 14 us for a 0.77 MB file where copying it costs a
   *     core 17) and the copy engine reads their scans where they lie, exactly like a `pinned` job's.
   *     input_cache_mb = 0 (default): a registration lives as long as the groups that use the buffer and is undone
   *       by the last of them (at most 4 GB held at a time; what does not fit is copied) — NOTHING of the caller's
   *       memory is registered once jga_pipeline_run() has returned,
   *       so the caller may free() or reuse its buffers as it pleases.  (Short runs on a host with cores to spare
   *       copy small files into the group's pinned blob instead: one copy call per group, see csrc/pipeline.cpp.)
   *     input_cache_mb > 0: a PERSISTENT cache of that many MB, keyed by address, least recently used out first:
   *       a buffer seen for the input_cache_sight-th time (0 = 1) is registered and STAYS registered after the run.
   *       Contract: a buffer the cache may hold must stay allocated until jga_pipeline_forget_input() has been
   *       called for it or the pipeline is destroyed — freeing registered memory is undefined in HIP.  As a second
   *       line of defence every entry carries a fingerprint of its file (size, first and last 64 bytes, sixteen
   *       words of the scan) that is re-read at every sight: a buffer handed out again at the same address with
   *       other contents loses its registration and is registered afresh.
   *     -1: no registration — the upload's copies name the callers' ordinary buffers and the runtime pins what they
   *       touch (per copy).  -2: every file that is not `pinned` is copied into the group's pinned blob by a host
   *       core (rounds 2-3). */
  int input_cache_mb;
  int input_cache_sight;
  int reserved_[6];            /* zero */
} jga_pipeline_config;

typedef struct jga_job {
  const unsigned char *jpeg;   /* in : file bytes (caller keeps alive) */
  int size;
  unsigned char *host_out;     /* in : destination if copy_back (may be NULL).  A job that FAILS (status 1) leaves it
                                * unspecified: copies are queued before the decode's verdict is known — a damaged file's
                                * buffer may hold partial pixels, then zeros (the same holds for img->pixels / plane data
                                * when HIPJPEG's decode_image returns EXIT_FAILURE) */
  unsigned char *dev_out;      /* in : device destination, or NULL = internal */
  int status;                  /* out: 0 ok, 1 failed */
  int width, height, nplanes;  /* out */
  long long h2d_bytes;         /* out: coefficient bytes this image sent over PCIe */
  int pinned;                  /* in : bit 0: `jpeg` lies in pinned memory (transport 2 with on-device
                                * unstuffing then uploads the scan straight from it); bit 1: `host_out`
                                * does (copy_back then writes the pixels straight into it: no staging
                                * buffer, no host memcpy) — jga_host_malloc_pinned / jga_host_register */
  long long host_bytes;        /* out: bytes of this file a host core read (copied or cleaned up) on the way to
                                * the device; 0 when the copy engine read the scan where it lies */
} jga_job;

/* Zero the configuration (every 0 is a documented default) and stamp it with this header's
 * struct sizes.  A macro on purpose: it must expand in the CALLER's translation unit. */
#define jga_pipeline_config_init(cfg) do { memset((cfg), 0, sizeof(jga_pipeline_config)); \
  (cfg)->struct_size = (int)sizeof(jga_pipeline_config); (cfg)->job_size = (int)sizeof(jga_job); } while (0)
jga_pipeline *jga_pipeline_create(const jga_pipeline_config *cfg);
/* Decode jobs[0..n) ; returns when all are complete (outputs valid).  One run at a time per
 * pipeline (its lanes and their threads — which live from create to destroy, in the process that
 * created them — belong to the run); several pipelines may run side by side. */
int  jga_pipeline_run(jga_pipeline *pl, jga_job *jobs, int n);
/* The groups a transport-2 pipeline of `lanes` lanes and `batch` 4K-frame equivalents per group (0: the
 * defaults) would cut jobs[0..n) into: group_of[i] = the group of job i; returns the number of groups.
 * Host logic only (no device is touched): same-geometry jobs in arrival order, groups sized by pixels,
 * short jobs cut finer, a long job's first groups rising in size. */
int  jga_pipeline_plan(int lanes, int batch, const jga_job *jobs, int n, int *group_of);
/* The same for a whole configuration (its depth and batch): exactly the plan jga_pipeline_run() of a pipeline
 * created from `cfg` makes. */
int  jga_pipeline_plan_cfg(const jga_pipeline_config *cfg, const jga_job *jobs, int n, int *group_of);
/* The persistent input cache (input_cache_mb > 0): register [jpeg, jpeg + size) now (a reader that fills its ingest
 * buffers before the first run) / drop it from the cache at once — REQUIRED before the buffer is freed or reused
 * for anything the device must not see.  Both return EXIT_SUCCESS or EXIT_FAILURE; forgetting a buffer the cache
 * does not hold is not an error; registering without a persistent cache is. */
int  jga_pipeline_register_input(jga_pipeline *pl, const unsigned char *jpeg, int size);
int  jga_pipeline_forget_input(jga_pipeline *pl, const unsigned char *jpeg);
/* Counters of the pipeline since it was created: [0] buffers registered, [1] MB registered now, [2] jobs whose
 * scan the copy engine read where it lay, [3] jobs whose scan a host core copied or cleaned up, [4] registrations
 * evicted, [5] microseconds spent in hipHostRegister, [6] where the scan clean-up of a long run takes place (0 host,
 * 1 device: what unstuff = 0 came to), [7] bytes of callers' files read by host cores, [8] registrations dropped
 * because the buffer's contents were no longer the file they were made for (the fingerprint check).  Fills at most n
 * entries, returns how many exist. */
int  jga_pipeline_counters(const jga_pipeline *pl, long long *out, int n);
void jga_pipeline_destroy(jga_pipeline *pl);

/* --- GPU entropy stage (BASELINE config 5): the scan is
 *     decoded ON THE GPU by self-synchronising parallel Huffman decoding —
 *     replaces the serial host loop src/xjpeg.c:449-632 (restart handling
 *     593-629) for callers that want no host Huffman work and 8x fewer PCIe
 *     bytes.  Works with or without DRI; restart intervals are extra hard sync
 *     points.  Output = the same QUANT-stage buffers as jga_entropy_decode().
 *     prepare(): host parses headers, uploads tables + compressed scan bytes
 *     (async on `stream`).  decode(): device only; callable repeatedly. ------ */
typedef struct jga_huff_batch jga_huff_batch;
jga_huff_batch *jga_huff_create(int max_images, long long max_scan_bytes);
void jga_huff_destroy(jga_huff_batch *b);
int jga_huff_prepare(jga_huff_batch *b, const unsigned char *const *jpegs,
 const int *sizes, int n, jga_geom *geom, void *stream);
int jga_huff_decode(jga_huff_batch *b, short *d_coef, long long coef_stride,
 void *stream);
/* The same decode for a caller that runs a block-decode kernel next: the planes' DC positions are
 * left holding the DC DIFFERENCES (src/xjpeg.c:480 not applied), the DC values arrive in d_dc —
 * image i at d_dc + i*dc_stride, one int16 per 128-byte slot of the image's coefficient buffer
 * (dc_stride >= coef_shorts/64) — for jga_idct_rgb_batch_dc / jga_idct_yuv_batch_dc.  Saves the
 * strided 2-byte pass over the planes that puts them in place. */
int jga_huff_decode_split(jga_huff_batch *b, short *d_coef, long long coef_stride,
 short *d_dc, long long dc_stride, void *stream);
/* jga_huff_decode_split in two halves, for a caller that wants ONE host wait per decode: _begin queues the decode
 * on `stream` and returns; the caller queues what consumes the planes (jga_idct_*_batch_dc, copies of its output)
 * on the same stream; _end waits for the stream, finishes the decode if its first burst of synchronisation rounds
 * had not settled it, and sets *valid_behind to 1 if the work queued in between saw the final planes and DC values,
 * 0 if it has to be queued again (streams that need more rounds than were queued up front: rare).  Returns and
 * verdicts as jga_huff_decode_split. */
int jga_huff_decode_split_begin(jga_huff_batch *b, short *d_coef, long long coef_stride,
 short *d_dc, long long dc_stride, void *stream);
int jga_huff_decode_split_end(jga_huff_batch *b, int *valid_behind);
/* Per-image outcome of the last jga_huff_prepare (also after it failed): 0 usable, 1 not
 * (damaged or unsupported file), 2 a valid file the device format cannot hold — Huffman
 * tables with too many long-code groups, or a frame beyond the kernels' 32-bit bit
 * positions / plane offsets (>= 4 GiB of planes): decode that one with jga_entropy_decode. */
int jga_huff_prepare_verdict(const jga_huff_batch *b, int i);
long long jga_huff_upload_bytes(const jga_huff_batch *b);
int jga_huff_last_rounds(const jga_huff_batch *b);
/* Subsequences the HOST walked in the last decode because the stream did not fall into step
 * on its own (periodic data: flat areas, letterbox bars); 0 for ordinary photographs. */
int jga_huff_last_assisted(const jga_huff_batch *b);
/* After a jga_huff_decode that returned EXIT_FAILURE because of damaged DATA: how many images
 * of the batch were affected (0: the failure was something else), and image i's verdict (0 ok,
 * bit 0 "entropy data ended early" / inconsistent stream, bit 1 coefficient index outside the
 * block).  The planes of the other images are complete and correct; a damaged image's planes are
 * UNDEFINED — whatever its lanes reached, over what the buffer held before (the decode clears nothing) —
 * and must not be handed on (jga_pipeline_run zeroes the output of such a job). */
int jga_huff_image_errors(const jga_huff_batch *b);
int jga_huff_image_error(const jga_huff_batch *b, int i);
const unsigned short *jga_huff_qtabs(const jga_huff_batch *b);
/* The same tables in device memory (they travel with prepare()'s upload): valid for work queued on prepare()'s
 * stream after it, until the next prepare(); NULL if nothing is prepared. */
const unsigned short *jga_huff_qtabs_device(const jga_huff_batch *b);
/* Where the scan's byte-level clean-up happens (stuffed zeros, fill bytes, RSTn markers ->
 * restart segments; T.81 B.1.1.5, the reference's bit reader src/xjpeg.c:113-127, 593-629):
 * 0 = on the host inside jga_huff_prepare (one core unstuffs ~12 GB/s), 1 = on the GPU — prepare
 * then only parses the marker segments and copies the raw scans into pinned memory.  Same
 * planes either way; with 1 a damaged restart structure is reported by jga_huff_decode
 * (per-image verdicts) instead of jga_huff_prepare.  Default 0, or JGA_HUFF_DEVICE_UNSTUFF. */
void jga_huff_set_device_unstuff(jga_huff_batch *b, int on);
/* The JPEG buffers given to jga_huff_prepare are pinned (jga_host_malloc_pinned /
 * jga_host_register): with on-device unstuffing their scans are copied by the DMA engine
 * straight from where they lie and the host touches no entropy-coded byte at all.  The buffers
 * must stay unchanged until the stream has executed prepare()'s copies. */
void jga_huff_set_inputs_pinned(jga_huff_batch *b, int on);
/* 1: the host waits inside jga_huff_decode poll the stream's event and sleep in between instead of
 * spinning in hipStreamSynchronize — for pipelines whose lanes outnumber the CPUs they may use
 * (a wait on a hipEventBlockingSync event spins just the same on this stack: csrc/host_wait.h). */
void jga_huff_set_blocking_waits(jga_huff_batch *b, int on);
/* prepare() queues its uploads on `copy_stream` (NULL = its own stream) and makes its own stream wait
 * for them: batches sharing a copy stream upload in the order they were prepared. */
void jga_huff_set_copy_stream(jga_huff_batch *b, void *copy_stream);
/* 1: other decodes run on the device at the same time (a pipeline's lanes).  A batch that would
 * leave most of an idle device empty takes the kernels with the shortest chain of steps; told
 * that the device is shared, it takes the ones that leave the most of it to the others; 2: ... and for a long
 * time (a pipeline's long run): nothing that shortens this batch's own chain at the others' expense. */
void jga_huff_set_device_shared(jga_huff_batch *b, int on);
/* `fn(arg, bytes, copies)` is called by prepare() right before it queues its upload of `bytes`
 * bytes in `copies` copy calls (a caller that runs several batches may want their uploads to cross
 * the link one after the other: the pipeline does).  NULL clears it. */
void jga_huff_set_upload_gate(jga_huff_batch *b, void (*fn)(void *arg, long long bytes, int copies), void *arg);
/* With a gate set: `poll(arg)` (the gate's arg) answers "would the gate let me through now?" — non-zero: yes, and the
 * turn is then the caller's (the gate call that follows must return at once).  prepare() with the clean-up on the
 * device asks between copying, one by one, the files it was told to read where they lie into its pinned blob
 * instead: a batch that waits for the link anyway then crosses it as ONE copy call at its full rate (55 GB/s; a
 * copy call per 0.8 MB file moves ~40 from two batches side by side).  Cleared with the gate. */
void jga_huff_set_upload_poll(jga_huff_batch *b, int (*poll)(void *arg));
/* 1: a host core read image i's scan in the last prepare() (copied it into the pinned blob, or cleaned it up),
 * 0: the copy engine read it where it lies, -1: no such image. */
int  jga_huff_image_copied(const jga_huff_batch *b, int i);
/* Returns when the last prepare()'s upload has arrived on the device. */
int  jga_huff_wait_upload(jga_huff_batch *b);
/* Host threads prepare() fans out over (0 = one per image, at most 64). */
void jga_huff_set_threads(jga_huff_batch *b, int nthreads);
/* Per-image version of jga_huff_set_inputs_pinned for the NEXT prepare(): flags[i] != 0 says image i's
 * scan is read where it lies by a copy call naming the buffer (pinned / registered memory; or ordinary memory, which
 * the runtime pins per copy) when the clean-up runs on the device, 0 that a host core copies it into the batch's
 * pinned blob.  NULL: back to the all-or-nothing setting.  The array is read during prepare() only. */
void jga_huff_set_input_flags(jga_huff_batch *b, const unsigned char *flags, int n);
/* Bytes of the callers' files the last prepare() read on the host (0 = all DMA'd in place). */
long long jga_huff_host_bytes(const jga_huff_batch *b);
/* Tunables of a batch object that used to be JGA_HUFF_* environment variables (still read by builds made
 * with -DJGA_TUNING).  value 0 = the default.  Returns EXIT_FAILURE for an unknown option or value. */
enum {
  JGA_HUFF_OPT_SUB_BYTES = 1,      /* subsequence length: 32, 64 or 128; 0 = by batch (256 / 512 were removed in round 6) */
  JGA_HUFF_OPT_ASSIST_AFTER = 2,   /* rounds before the host walks unsettled stretches (default 12) */
  JGA_HUFF_OPT_SPECULATE = 3,      /* 0 / 1 = the tail is queued behind the first rounds, -1 = never */
  /* 4: retired (round 4's JGA_HUFF_OPT_PIECES): rejected, so that a stale caller does not switch something else on */
  JGA_HUFF_OPT_TRACE = 5           /* 1: what prepare() and decode() spent where, on stderr */
};
int jga_huff_set_option(jga_huff_batch *b, int option, int value);

/* --- the plugin's settings (HIPJPEG_DECODE_CTX_VTBL has no room for options: decode_alloc takes a
 *     jpeg_info and nothing else, src/jpeg_wrap.h:45).  Process-wide, read by decoder contexts when they
 *     are allocated; what JGA_PLUGIN_* environment variables used to say. */
typedef struct jga_plugin_config {
  int struct_size;             /* sizeof(jga_plugin_config) of the caller's header */
  int register_buffers;        /* how the caller's image and file buffers meet the device:
                                *  0 (default, any caller): the copies name the caller's ordinary memory and the
                                *    runtime pins what they touch (and caches that) — pixels arrive at link speed,
                                *    big files are DMA'd where they lie, no host core passes over either;
                                *  1: the plugin registers the buffers for the life of the decoder context
                                *    (hipHostRegister) — for callers that keep image and file in place for that long
                                *    (the reference's main loop does, src/jpeg_gpu.c:612-613, 1231-1237);
                                * -1: copies staged through the context's pinned buffers (rounds 2-3's default) */
  int host_entropy;            /* 1: Huffman decoding on the host (csrc/entropy.c) for YUV / RGB too */
  int copy_team;               /* staged copy back of frames of 12 MB and more: 0 = two helper threads move the
                                * pieces (default), -1 = the calling thread alone */
  int reserved_;               /* zero */
} jga_plugin_config;
#define jga_plugin_config_init(cfg) do { memset((cfg), 0, sizeof(jga_plugin_config)); \
  (cfg)->struct_size = (int)sizeof(jga_plugin_config); } while (0)
int jga_plugin_configure(const jga_plugin_config *cfg);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* JPEG_GPU_AMD_H */
