// wn_host.h -- host-side helpers shared by the translation units behind the C ABI (include/walnuts_hip.h):
// the error object, HIP status checks, exception -> error-code mapping, a device buffer.
#ifndef WN_HOST_H
#define WN_HOST_H
#include "wn_hip.h"

#include <sstream>
#include <stdexcept>
#include <string>

#include "../../include/walnuts_hip.h"

// ---- errors (python/src/walnutpie/errors.hpp:10-72) --------------------------------------
struct WalnutpyError {
  std::string msg;
  WalnutpyErrorType type;
};

namespace {

void hip_check(hipError_t e, const char* what) {
  if (e != hipSuccess) {
    std::stringstream ss;
    ss << "HIP error in " << what << ": " << hipGetErrorString(e);
    throw std::runtime_error(ss.str());
  }
}
#define HIP_OK(expr) hip_check((expr), #expr)

template <class F>
int guarded(WalnutpyError** err, F f) {
  try {
    f();
    return 0;
  } catch (const std::invalid_argument& e) {
    if (err) *err = new WalnutpyError{e.what(), config};
  } catch (const std::exception& e) {
    if (err) *err = new WalnutpyError{e.what(), generic};
  } catch (...) {
    if (err) *err = new WalnutpyError{"Unknown error", generic};
  }
  return -1;
}

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  void alloc(size_t count) {
    release();
    n = count;
    if (count) HIP_OK(hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)));
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  ~DevBuf() { release(); }
};

}  // namespace

#endif  // WN_HOST_H
