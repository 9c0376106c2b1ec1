#define _GNU_SOURCE
/* layout.c — frame geometry and `image` buffers for the decode path.
 *
 * Host-side mirror of the reference's data model:
 *   jga_image_init / _zero / _clear  <->  image_init / image_zero / image_clear
 *                                         (reference src/image.c:24-124)
 *   jga_geom_from_header             <->  the layout image_init derives
 *                                         (src/image.c:49-70, 86-95)
 *   jga_block_offset                 <->  block placement, src/xjpeg.c:556-561
 * Behaviour (field values, buffer sizes, 16-byte alignment) is the
 * reference's; the code is written against SURVEY.md Appendix B.
 */
#include <stdarg.h>
#include <stdio.h>
#include <sched.h>
#include <stdlib.h>
#include <unistd.h>
#include <string.h>
#include "jga_internal.h"

/* ---- ABI checks: x86-64 SysV numbers from SURVEY.md §8b ----------------- */
#define JGA_SA(c) _Static_assert(c, #c)
JGA_SA(sizeof(image_plane) == 56);
JGA_SA(sizeof(image) == 208);
JGA_SA(sizeof(jpeg_quant) == 136);
JGA_SA(sizeof(jpeg_component) == 24);
JGA_SA(sizeof(jpeg_header) == 640);
JGA_SA(sizeof(jpeg_info) == 16);
JGA_SA(sizeof(jpeg_decode_ctx_vtbl) == 40);
JGA_SA(offsetof(image_plane, xdec) == 4);
JGA_SA(offsetof(image_plane, xstride) == 8);
JGA_SA(offsetof(image_plane, ystride) == 12);
JGA_SA(offsetof(image_plane, width) == 16);
JGA_SA(offsetof(image_plane, height) == 18);
JGA_SA(offsetof(image_plane, data) == 24);
JGA_SA(offsetof(image_plane, coef) == 32);
JGA_SA(offsetof(image_plane, cstride) == 40);
JGA_SA(offsetof(image_plane, packed) == 44);
JGA_SA(offsetof(image_plane, index) == 48);
JGA_SA(offsetof(image, nplanes) == 4);
JGA_SA(offsetof(image, plane) == 8);
JGA_SA(offsetof(image, coef) == 176);
JGA_SA(offsetof(image, packed) == 184);
JGA_SA(offsetof(image, index) == 192);
JGA_SA(offsetof(image, pixels) == 200);
JGA_SA(offsetof(jpeg_header, subsamp) == 16);
JGA_SA(offsetof(jpeg_header, restart_interval) == 20);
JGA_SA(offsetof(jpeg_header, comp) == 24);
JGA_SA(offsetof(jpeg_header, quant) == 96);
JGA_SA(offsetof(jpeg_info, buf) == 8);

/* ---- error reporting: one line on stderr + retrievable string ----------- */
static __thread char jga_err[256];

int jga_fail(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(jga_err, sizeof(jga_err), fmt, ap);
  va_end(ap);
  {
    const char *quiet = getenv("JGA_QUIET");          /* unset, empty or "0": speak */
    if (!quiet || !*quiet || *quiet == '0') fprintf(stderr, "%s\n", jga_err);
  }
  return EXIT_FAILURE;
}

const char *jga_last_error(void) { return jga_err; }

/* A/B knobs (jga_tune.h): an environment variable in the tuning build, nothing in the product. */
const char *jga_tune(const char *name) {
#ifdef JGA_TUNING
  return getenv(name);
#else
  (void)name;
  return NULL;
#endif
}
const char *jga_version(void) { return "jpeg_gpu_amd 0.1 (gfx950)"; }

/* quota / period of one cgroup directory: cgroup v2 "cpu.max" ("quota period" or "max period"),
 * v1 "cpu.cfs_quota_us" + "cpu.cfs_period_us" (-1 = unlimited).  Returns CPUs' worth of run time,
 * 0 when the directory sets no limit or cannot be read. */
static double cgroup_dir_quota(const char *dir) {
  char path[512 + 32];                 /* (dir is at most 512 bytes: room for the longest file name) */
  double quota = 0, period = 0;
  FILE *f;
  if (strlen(dir) >= 512) return 0.0;
  snprintf(path, sizeof(path), "%s/cpu.max", dir);
  f = fopen(path, "r");
  if (f) {
    const int got = fscanf(f, "%lf %lf", &quota, &period);      /* "max ..." scans nothing */
    fclose(f);
    return (got == 2 && quota > 0 && period > 0) ? quota/period : 0.0;
  }
  snprintf(path, sizeof(path), "%s/cpu.cfs_quota_us", dir);
  f = fopen(path, "r");
  if (!f) return 0.0;
  if (fscanf(f, "%lf", &quota) != 1) quota = 0;
  fclose(f);
  snprintf(path, sizeof(path), "%s/cpu.cfs_period_us", dir);
  f = fopen(path, "r");
  if (!f) return 0.0;
  if (fscanf(f, "%lf", &period) != 1) period = 0;
  fclose(f);
  return (quota > 0 && period > 0) ? quota/period : 0.0;
}

/* The tightest CPU quota that applies to this process: its own cgroup (from /proc/self/cgroup:
 * "0::/path" on v2, "N:cpu,cpuacct:/path" on v1) and every ancestor up to the mount point —
 * a limit anywhere on the way up binds.  0 = none found. */
static double cgroup_cpu_quota(void) {
  static const char *roots[2] = {"/sys/fs/cgroup", "/sys/fs/cgroup/cpu"};
  char line[512], rel[2][400] = {"", ""};
  double best = 0.0;
  int k;
  FILE *f = fopen("/proc/self/cgroup", "r");
  if (f) {
    while (fgets(line, sizeof(line), f)) {
      char *c1 = strchr(line, ':'), *c2 = c1 ? strchr(c1 + 1, ':') : NULL, *nl;
      if (!c2) continue;
      nl = strchr(c2, '\n');
      if (nl) *nl = 0;
      *c2 = 0;
      if (c1[1] == 0) snprintf(rel[0], sizeof(rel[0]), "%s", c2 + 1);                    /* v2: "0::/path" */
      else if (strstr(c1 + 1, "cpu") && !strstr(c1 + 1, "cpuset")) snprintf(rel[1], sizeof(rel[1]), "%s", c2 + 1);
    }
    fclose(f);
  }
  for (k = 0; k < 2; k++) {
    char dir[512];
    snprintf(dir, sizeof(dir), "%s%s", roots[k], strcmp(rel[k], "/") ? rel[k] : "");
    for (;;) {
      const double q = cgroup_dir_quota(dir);
      char *slash;
      if (q > 0 && (best == 0.0 || q < best)) best = q;
      if (strlen(dir) <= strlen(roots[k])) break;
      slash = strrchr(dir, '/');
      if (!slash) break;
      *slash = 0;
    }
  }
  return best;
}

/* CPUs this process can really keep busy: the affinity mask, cut down to the container's
 * cgroup grant (a box may show 256 CPUs and grant 16, and threads beyond the grant only get
 * the whole group throttled).  Default thread counts come from here, never from the raw CPU
 * count.  JGA_CPU_BUDGET overrides (a launcher that knows the rank's share passes it on). */
int jga_cpu_budget(void) {
  cpu_set_t set;
  int n = 0;
  const char *e = getenv("JGA_CPU_BUDGET");
  if (e && atoi(e) > 0) return atoi(e);
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
  if (n < 1) n = (int)sysconf(_SC_NPROCESSORS_ONLN);
  if (n < 1) n = 1;
  {
    const double q = cgroup_cpu_quota();
    const int grant = (int)(q + 0.5);
    if (q > 0 && grant >= 1 && grant < n) n = grant;
    else if (q > 0 && grant < 1) n = 1;
  }
  return n;
}

/* number of bits needed to represent v (glj_ilog semantics) */
int jga_ilog(unsigned v) {
  int n = 0;
  while (v) { n++; v >>= 1; }
  return n;
}

int jga_subsamp_of(int xdec, int ydec, int ncomps) {
  if (ncomps == 1) return JPEG_SUBSAMP_MONO;
  if (xdec == 0 && ydec == 0) return JPEG_SUBSAMP_444;
  if (xdec == 1 && ydec == 0) return JPEG_SUBSAMP_422;
  if (xdec == 1 && ydec == 1) return JPEG_SUBSAMP_420;
  if (xdec == 0 && ydec == 1) return JPEG_SUBSAMP_440;
  if (xdec == 2 && ydec == 0) return JPEG_SUBSAMP_411;
  return JPEG_SUBSAMP_UNKNOWN;
}

int jga_geom_from_header(jga_geom *g, const jpeg_header *h) {
  int i, hmax = 0, vmax = 0;
  long long coef = 0, data = 0, blocks = 0;
  memset(g, 0, sizeof(*g));
  if (h->ncomps != 1 && h->ncomps != 3) {
    return jga_fail("Unsupported number of components %i", h->ncomps);
  }
  for (i = 0; i < h->ncomps; i++) {
    const jpeg_component *c = &h->comp[i];
    if (c->hsamp < 1 || c->vsamp < 1 || c->hblocks < 1 || c->vblocks < 1) {
      return jga_fail("Invalid component %i in header", i);
    }
    if (c->hsamp > hmax) hmax = c->hsamp;
    if (c->vsamp > vmax) vmax = c->vsamp;
  }
  g->width = h->width;
  g->height = h->height;
  g->nplanes = h->ncomps;
  g->restart_interval = h->restart_interval;
  g->nhmb = h->comp[0].hblocks/h->comp[0].hsamp;
  g->nvmb = h->comp[0].vblocks/h->comp[0].vsamp;
  g->w0 = h->comp[0].hblocks*8;
  for (i = 0; i < h->ncomps; i++) {
    const jpeg_component *c = &h->comp[i];
    jga_plane_geom *p = &g->plane[i];
    p->hblocks = c->hblocks;
    p->vblocks = c->vblocks;
    p->xdec = jga_ilog(hmax) - jga_ilog(c->hsamp);
    p->ydec = jga_ilog(vmax) - jga_ilog(c->vsamp);
    p->cstride = (c->vblocks + ((1 << p->xdec) - 1)) >> p->xdec;
    p->qidx = i;
    p->coef_off = coef;
    p->data_off = data;
    /* The packing assumes every plane fills a luma-width row of blocks once
       scaled by its horizontal decimation (true whenever luma carries the
       largest sampling factors, the only case the reference handles). */
    if ((c->hblocks << p->xdec) != h->comp[0].hblocks) {
      return jga_fail("Unsupported sampling: plane %i is not luma width >> xdec", i);
    }
    coef += ((long long)c->hblocks << (p->xdec + 6))*p->cstride;
    data += (long long)c->hblocks*c->vblocks*64;
    blocks += (long long)c->hblocks*c->vblocks;
  }
  g->coef_shorts = coef;
  g->coef_blocks = blocks;
  g->yuv_bytes = data;
  g->rgb_bytes = (long long)h->width*h->height*h->ncomps;
  g->subsamp = h->ncomps == 1 ? JPEG_SUBSAMP_MONO
   : jga_subsamp_of(g->plane[1].xdec, g->plane[1].ydec, 3);
  if (h->ncomps == 3 && (g->plane[1].xdec != g->plane[2].xdec
   || g->plane[1].ydec != g->plane[2].ydec || g->plane[0].xdec
   || g->plane[0].ydec)) {
    g->subsamp = JPEG_SUBSAMP_UNKNOWN;
  }
  return EXIT_SUCCESS;
}

long long jga_block_offset(const jga_geom *g, int p, int bx, int by) {
  const jga_plane_geom *pl = &g->plane[p];
  long long rs = (long long)g->w0 << 3;
  return pl->coef_off + rs*(by >> pl->xdec)
   + (rs >> pl->xdec)*(by & ((1 << pl->xdec) - 1)) + ((long long)bx << 6);
}

/* ---- image buffers ------------------------------------------------------ */
#define JGA_IMAGE_ALIGN (16)

static void *aligned_alloc16(size_t bytes) {
  void *p = NULL;
  if (posix_memalign(&p, JGA_IMAGE_ALIGN, bytes ? bytes : JGA_IMAGE_ALIGN)) {
    return NULL;
  }
  return p;
}

int jga_image_init(image *img, jpeg_header *header) {
  jga_geom g;
  int i;
  long long index_ints = 0;
  memset(img, 0, sizeof(*img));
  if (jga_geom_from_header(&g, header) != EXIT_SUCCESS) return EXIT_FAILURE;
  if (g.w0 > 65535 || g.plane[0].vblocks*8 > 65535) {
    return jga_fail("Image too large for the 16-bit plane dimensions");
  }
  img->width = (unsigned short)header->width;
  img->height = (unsigned short)header->height;
  img->nplanes = header->ncomps;
  for (i = 0; i < img->nplanes; i++) {
    image_plane *pl = &img->plane[i];
    pl->width = (unsigned short)(g.plane[i].hblocks << 3);
    pl->height = (unsigned short)(g.plane[i].vblocks << 3);
    pl->xstride = 1;
    pl->ystride = pl->width;
    pl->xdec = (unsigned char)g.plane[i].xdec;
    pl->ydec = (unsigned char)g.plane[i].ydec;
    pl->cstride = g.plane[i].cstride;
    pl->data = (unsigned char *)aligned_alloc16((size_t)pl->ystride*pl->height);
    if (!pl->data) goto oom;
    index_ints += (long long)(g.plane[i].hblocks << pl->xdec)*pl->cstride;
  }
  img->pixels = (unsigned char *)aligned_alloc16((size_t)img->width*img->height*3);
  img->coef = (short *)aligned_alloc16((size_t)g.coef_shorts*sizeof(short));
  img->index = (int *)aligned_alloc16((size_t)index_ints*sizeof(int));
  if (!img->pixels || !img->coef || !img->index) goto oom;
  {
    int *index = img->index;
    for (i = 0; i < img->nplanes; i++) {
      image_plane *pl = &img->plane[i];
      pl->coef = img->coef + g.plane[i].coef_off;
      pl->index = index;
      index += (g.plane[i].hblocks << pl->xdec)*pl->cstride;
    }
  }
  return EXIT_SUCCESS;
oom:
  jga_image_clear(img);
  return jga_fail("Out of memory allocating image buffers");
}

void jga_image_zero(image *img) {
  int i;
  long long blocks = 0;
  for (i = 0; i < img->nplanes; i++) {
    image_plane *pl = &img->plane[i];
    memset(pl->data, 0, (size_t)pl->ystride*pl->height);
    blocks += (long long)((pl->width >> 3) << pl->xdec)*pl->cstride;
    pl->packed = 0;
  }
  memset(img->pixels, 0, (size_t)img->width*img->height*3);
  memset(img->coef, 0, (size_t)blocks*64*sizeof(short));
  memset(img->index, 0, (size_t)blocks*sizeof(int));
  img->packed = 0;
}

void jga_image_clear(image *img) {
  int i;
  for (i = 0; i < img->nplanes && i < NPLANES_MAX; i++) free(img->plane[i].data);
  free(img->pixels);
  free(img->coef);
  free(img->index);
  memset(img, 0, sizeof(*img));
}
