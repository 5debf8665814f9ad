/*
 * mfn_ref.c -- CPU oracle for the MaskFlownet matching hot path (TEST INFRASTRUCTURE ONLY).
 * See mfn_ref.h: restatement of the MXNet 1.5.x CPU operators behind
 * /root/reference/network/layer.py and network/MaskFlownet.py:195,441.  PARITY UNPINNED
 * against MXNet itself (MXNet is not installable here and the reference has no tests).
 * Build: `make -C oracle` (gcc -O2 -ffp-contract=off: no FMA contraction, like a generic
 * x86-64 MXNet wheel).
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include "mfn_ref.h"

/* DeformableConvolution: where the bilinear fractions come from.
 *   0 (default)  MXNet's kernel as written -- deformable_im2col.h interpolates on the (h_in, w_in)-relative
 *                map_h = i*dilation + offset inside the cur_height x cur_width window (what the HIP kernels follow);
 *   1            SURVEY.md Appendix A.3's statement -- fractions from the absolute h_im = h_in + i*dilation + offset.
 * Mathematically the same sample; in fp32 the two differ by ~1e-6 relative (h_im carries the rounding of one more
 * addition).  Only a real MXNet can say which one its binary does; the flag keeps the alternative one call away
 * (SURVEY.md A.5). */
static int g_dc_fraction_mode = 0;
void mfn_ref_set_dc_fraction_mode(int mode) { g_dc_fraction_mode = mode ? 1 : 0; }
int mfn_ref_get_dc_fraction_mode(void) { return g_dc_fraction_mode; }

#define MFN_CAT_(a, b) a##b
#define MFN_CAT(a, b) MFN_CAT_(a, b)

/* fp32, faithful */
#define REAL float
#define FN(name) MFN_CAT(mfn_ref_, name)
#define RABS(x) fabsf(x)
#include "mfn_ref_body.inc"
#undef REAL
#undef FN
#undef RABS

/* fp64 arbiter */
#define REAL double
#define FN(name) MFN_CAT(mfn_ref64_, name)
#define RABS(x) fabs(x)
#include "mfn_ref_body.inc"
#undef REAL
#undef FN
#undef RABS

const char *mfn_ref_version(void) { return "mfn_ref 0.1 (restates MXNet 1.5.x CPU ops; parity unpinned)"; }
