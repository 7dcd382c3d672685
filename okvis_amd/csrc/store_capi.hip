// C-ABI of the host-side window container (include/okvis_amd_ba.h, "incremental structure updates"): host code only, compiled
// into the same shared library as the solver so that okvis_ba_patch_window and okvis_ba_store_patch are one implementation
// (ba_store.hpp).
#include <new>

#include "ba_store.hpp"

struct okvis_ba_window_store {
  ba::WindowStore st;
};

extern "C" {

int okvis_ba_store_create(const okvis_ba_window* w, okvis_ba_window_store** out) {
  if (!w || !out) return OKVIS_BA_ERR_ARG;
  *out = nullptr;
  okvis_ba_window_store* s = new (std::nothrow) okvis_ba_window_store;
  if (!s) return OKVIS_BA_ERR_ARG;
  int rc = OKVIS_BA_ERR_ARG;
  try {
    rc = s->st.assign(*w);
  } catch (const std::bad_alloc&) {
    rc = OKVIS_BA_ERR_ARG;
  }
  if (rc != OKVIS_BA_OK) {
    delete s;
    return rc;
  }
  *out = s;
  return OKVIS_BA_OK;
}

int okvis_ba_store_patch(okvis_ba_window_store* s, const okvis_ba_patch* p) {
  if (!s || !p) return OKVIS_BA_ERR_ARG;
  try {
    return s->st.apply(*p);
  } catch (const std::bad_alloc&) {
    return OKVIS_BA_ERR_ARG;
  }
}

int okvis_ba_store_view(const okvis_ba_window_store* s, okvis_ba_window* out) {
  if (!s || !out) return OKVIS_BA_ERR_ARG;
  s->st.view(out);
  return OKVIS_BA_OK;
}

void okvis_ba_store_destroy(okvis_ba_window_store* s) { delete s; }

}  // extern "C"
