/* jga_tune.h — A/B knobs.
 *
 * The product reads its settings from jga_pipeline_config / jga_plugin_config / jga_huff_set_option.
 * The JGA_* environment variables the measurements of rounds 1-3 were taken with (and the tests of
 * alternate code paths still use) are read through jga_tune(), which is `return NULL` in
 * libjpeg_gpu_amd.so and `return getenv(name)` only in libjpeg_gpu_amd_tuning.so — the same objects
 * with layout.c compiled -DJGA_TUNING (jpeg_gpu_amd/build.py builds both; tests and tools/ select the
 * second with JGA_LIB_PATH).  Every site treats NULL as "not set".  What the default build still reads
 * from the environment: JGA_QUIET and JGA_CPU_BUDGET (layout.c), JGA_LIBJPEG (libjpeg_vtbl.c). */
#ifndef JGA_TUNE_H
#define JGA_TUNE_H (1)
#ifdef __cplusplus
extern "C"
#endif
const char *jga_tune(const char *name);
#endif
