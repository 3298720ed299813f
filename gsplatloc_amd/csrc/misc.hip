// Library identification and status strings.
#include "gsloc_common.h"

extern "C" const char* gsl_version(void) { return "gsloc_hip 0.1.0 gfx950"; }

extern "C" const char* gsl_status_string(int status) {
  switch (status) {
    case GSL_OK: return "ok";
    case GSL_ERR_BAD_ARG: return "bad argument";
    case GSL_ERR_WORKSPACE: return "workspace too small";
    case GSL_ERR_HIP: return "HIP launch error";
    default: return "unknown status";
  }
}
