/* v21_types.h -- plain-data types of the C ABI that device code shares with the host (included by v21.h; also embedded,
 * with csrc/fused_fwd.h, in the sources csrc/jit.hip compiles at run time: no #include of its own). */
#ifndef V21_TYPES_H
#define V21_TYPES_H
/* statistics of preprocess.par_transform (preprocess.py:49-110); the arithmetic is described at its use in v21.h */
typedef struct {
  int n;            /* parameter columns, <= 8 */
  int log_mask[8];  /* 1: the column is emulated in log10 (preprocess.py:77-78) */
  double zero_floor[8]; /* > 0: x == 0 is replaced by this first (fx == 0 -> 1e-6, preprocess.py:76) */
  double lo[8];     /* column minimum of the log-transformed TRAINING parameters (preprocess.py:100-101) */
  double span[8];   /* maximum - minimum (preprocess.py:106) */
} v21_affine_in;
#endif /* V21_TYPES_H */
