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
