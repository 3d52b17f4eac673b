// ABI version and error strings of libdrba_hip.so.
#include "common.hpp"

extern "C" {

int drba_abi_version(void) { return 1; }

const char *drba_error_string(int code) {
  switch (code) {
    case DRBA_OK: return "ok";
    case DRBA_EINVAL: return "invalid argument";
    case DRBA_EUNSUPPORTED: return "unsupported shape or configuration";
    case DRBA_ELAUNCH: return "HIP launch failure";
  }
  return "unknown error";
}

}  // extern "C"
