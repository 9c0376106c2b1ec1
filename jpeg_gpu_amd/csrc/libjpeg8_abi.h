/* libjpeg8_abi.h — the slice of the IJG libjpeg binary interface (ABI level 8, i.e. what
 * `libjpeg.so.8` from libjpeg-turbo exports) that csrc/libjpeg_vtbl.c talks to.
 *
 * The image has the runtime library but not <jpeglib.h>, so the few records
 * the comparison backend touches are declared here, by hand, for this one ABI level on
 * x86-64 SysV; every function is looked up with dlsym, nothing links against libjpeg.
 * Two guards keep a wrong declaration from going unnoticed: jpeg_CreateDecompress() rejects
 * a caller whose idea of sizeof(struct jpeg_decompress_struct) differs from the library's,
 * and tests/test_libjpeg_backend.py checks the decoded coefficients against the oracle.
 *
 * Only the decompression side is described; members this backend never reads are kept for
 * their size and position alone. */
#ifndef JGA_LIBJPEG8_ABI_H
#define JGA_LIBJPEG8_ABI_H (1)

#include <stddef.h>

#define LJ8_LIB_VERSION (80)
#define LJ8_HEADER_OK (1)            /* jpeg_read_header: a frame header was found */

typedef int lj8_bool;
typedef unsigned int lj8_dim;
typedef unsigned char *lj8_row;      /* one row of samples */
typedef lj8_row *lj8_rows;           /* rows of one component */
typedef lj8_rows *lj8_planes;        /* per component, for raw output */
typedef short lj8_block[64];         /* one block of levels, natural order */
typedef lj8_block *lj8_block_row;
typedef lj8_block_row *lj8_block_rows;
typedef struct lj8_virt_blocks *lj8_virt_blocks_ptr;   /* opaque */

enum { LJ8_CS_UNKNOWN, LJ8_CS_GRAYSCALE, LJ8_CS_RGB, LJ8_CS_YCBCR };
enum { LJ8_DCT_ISLOW, LJ8_DCT_IFAST, LJ8_DCT_FLOAT };

struct lj8_common;                   /* the leading members of the (de)compress records */

typedef struct lj8_error_mgr {
  void (*error_exit)(struct lj8_common *);
  void (*emit_message)(struct lj8_common *, int level);
  void (*output_message)(struct lj8_common *);
  void (*format_message)(struct lj8_common *, char *buffer);   /* buffer: 200 bytes */
  void (*reset_error_mgr)(struct lj8_common *);
  int msg_code;
  union { int i[8]; char s[80]; } msg_parm;
  int trace_level;
  long num_warnings;
  const char *const *message_table;
  int last_message;
  const char *const *addon_table;
  int first_addon, last_addon;
} lj8_error_mgr;

typedef struct lj8_memory_mgr {
  void *alloc_small, *alloc_large, *alloc_sarray, *alloc_barray;
  void *request_virt_sarray, *request_virt_barray, *realize_virt_arrays;
  void *access_virt_sarray;
  lj8_block_rows (*access_virt_barray)(struct lj8_common *, lj8_virt_blocks_ptr,
   lj8_dim first_row, lj8_dim rows, lj8_bool writable);
  void *free_pool, *self_destruct;
  long max_memory_to_use, max_alloc_chunk;
} lj8_memory_mgr;

typedef struct lj8_quant_tbl {
  unsigned short quantval[64];       /* natural order once read by the library */
  lj8_bool sent_table;
} lj8_quant_tbl;

typedef struct lj8_component {
  int id, index;
  int h_samp, v_samp;
  int quant_tbl_no, dc_tbl_no, ac_tbl_no;
  lj8_dim width_in_blocks, height_in_blocks;
  int dct_h_scaled, dct_v_scaled;
  lj8_dim downsampled_width, downsampled_height;
  lj8_bool needed;
  int mcu_width, mcu_height, mcu_blocks, mcu_sample_width;
  int last_col_width, last_row_height;
  lj8_quant_tbl *quant_table;
  void *dct_table;
} lj8_component;

typedef struct lj8_decompress {
  /* common to both directions */
  lj8_error_mgr *err;
  lj8_memory_mgr *mem;
  void *progress;
  void *client_data;
  lj8_bool is_decompressor;
  int global_state;
  /* source + what the header said */
  void *src;
  lj8_dim image_width, image_height;
  int num_components;
  int jpeg_color_space;
  /* decoding parameters the caller may set before jpeg_start_decompress */
  int out_color_space;
  unsigned scale_num, scale_denom;
  double output_gamma;
  lj8_bool buffered_image;
  lj8_bool raw_data_out;
  int dct_method;
  lj8_bool do_fancy_upsampling;
  lj8_bool do_block_smoothing;
  lj8_bool quantize_colors;
  int dither_mode;
  lj8_bool two_pass_quantize;
  int desired_number_of_colors;
  lj8_bool enable_1pass_quant, enable_external_quant, enable_2pass_quant;
  /* output description */
  lj8_dim output_width, output_height;
  int out_color_components, output_components;
  int rec_outbuf_height;
  int actual_number_of_colors;
  lj8_rows colormap;
  lj8_dim output_scanline;
  int input_scan_number;
  lj8_dim input_imcu_row;
  int output_scan_number;
  lj8_dim output_imcu_row;
  int (*coef_bits)[64];
  /* tables and frame description */
  lj8_quant_tbl *quant_tbl_ptrs[4];
  void *dc_huff_tbl_ptrs[4];
  void *ac_huff_tbl_ptrs[4];
  int data_precision;
  lj8_component *comp_info;
  lj8_bool is_baseline;
  lj8_bool progressive_mode;
  lj8_bool arith_code;
  unsigned char arith_dc_l[16], arith_dc_u[16], arith_ac_k[16];
  unsigned restart_interval;
  lj8_bool saw_jfif_marker;
  unsigned char jfif_major, jfif_minor, density_unit;
  unsigned short x_density, y_density;
  lj8_bool saw_adobe_marker;
  unsigned char adobe_transform;
  lj8_bool ccir601_sampling;
  void *marker_list;
  int max_h_samp, max_v_samp;
  int min_dct_h_scaled, min_dct_v_scaled;
  lj8_dim total_imcu_rows;
  unsigned char *sample_range_limit;
  /* current scan */
  int comps_in_scan;
  lj8_component *cur_comp_info[4];
  lj8_dim mcus_per_row, mcu_rows_in_scan;
  int blocks_in_mcu;
  int mcu_membership[10];
  int ss, se, ah, al;
  int block_size;
  const int *natural_order;
  int lim_se;
  int unread_marker;
  /* the library's private modules */
  void *master, *main_ctl, *coef, *post, *inputctl, *marker, *entropy, *idct, *upsample,
   *cconvert, *cquantize;
} lj8_decompress;

/* entry points (all resolved with dlsym) */
typedef lj8_error_mgr *(*lj8_std_error_fn)(lj8_error_mgr *);
typedef void (*lj8_create_decompress_fn)(lj8_decompress *, int version, size_t structsize);
typedef void (*lj8_mem_src_fn)(lj8_decompress *, const unsigned char *, unsigned long);
typedef int (*lj8_read_header_fn)(lj8_decompress *, lj8_bool require_image);
typedef lj8_virt_blocks_ptr *(*lj8_read_coefficients_fn)(lj8_decompress *);
typedef lj8_bool (*lj8_start_decompress_fn)(lj8_decompress *);
typedef lj8_dim (*lj8_read_raw_data_fn)(lj8_decompress *, lj8_planes, lj8_dim max_lines);
typedef lj8_dim (*lj8_read_scanlines_fn)(lj8_decompress *, lj8_rows, lj8_dim max_lines);
typedef lj8_bool (*lj8_finish_decompress_fn)(lj8_decompress *);
typedef void (*lj8_destroy_decompress_fn)(lj8_decompress *);

#endif
