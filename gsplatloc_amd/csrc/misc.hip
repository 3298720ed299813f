// Library identification and status strings.
#include "gsloc_common.h"

extern "C" const char* gsl_version(void) { return "gsloc_hip 0.1.0 gfx950"; }

// 1: Q0/Q1/Q2 are three arrays of float4; 4: they are columns of one array of 64-byte rows (build variant).
extern "C" int gsl_record_stride(void) { return GSL_QS; }

extern "C" const char* gsl_status_string(int status) {
  switch (status) {
    case GSL_OK: return "ok";
    case GSL_ERR_BAD_ARG: return "bad argument";
    case GSL_ERR_WORKSPACE: return "workspace too small";
    case GSL_ERR_HIP: return "HIP launch error";
    default: return "unknown status";
  }
}
