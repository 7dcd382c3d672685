// Host-side container of one window with O(edit) structure updates: what okvis::ceres::Map is to the reference
// (addParameterBlock / addResidualBlock / removeResidualBlock / removeParameterBlock, okvis_ceres/src/Map.cpp:292-565),
// as flat arrays that okvis_ba_upload can take as they are.  No HIP in here: okvis_ba_store_* works on a host without a GPU,
// okvis_ba_patch_window (ba_capi.hip) applies the same edit to the copy a patchable solver keeps of every uploaded window.
//
// Semantics of apply(patch), in this order:
//   1. removals by index into the window as it stands.  Removing a parameter block removes every term attached to it, like
//      Map::removeParameterBlock (Map.cpp:352-379): observations of a removed landmark / pose / extrinsics block, IMU terms and
//      priors on a removed block.  The dense marginalisation prior cannot lose a block: it has to be replaced in the same patch.
//   2. what is left is renumbered by stable compaction (relative order kept);
//   3. appended blocks take the next indices; appended terms and replaced prior families use the NEW numbering;
//   4. observations stay sorted by (landmark, pose, camera): appended ones are merged in (behind equal keys);
//   5. sparse value updates (new numbering).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/okvis_amd_ba.h"

namespace ba {

struct WindowStore {
  // parameter blocks
  std::vector<double> pose, sb, lm;
  std::vector<uint8_t> pose_fixed, sb_fixed;
  // cameras
  std::vector<double> cam_intr;
  std::vector<int32_t> cam_model;
  // observations
  std::vector<int32_t> obs_lm, obs_pose, obs_ext, obs_cam;
  std::vector<double> obs_uv, obs_sqrtw;
  double cauchy_b = 0.0;
  // IMU terms: every term owns its samples (flattened by view())
  struct Imu {
    int32_t pose0, sb0, pose1, sb1;
    int64_t t0, t1;
    std::vector<int64_t> s_t;
    std::vector<double> s_gyr, s_acc;
    double sb_ref[9];
    uint8_t ref_valid;   // 0 none / 1 the reference bias / 2 the whole preintegration record (okvis_ba_window::imu_sb_ref_valid)
    std::vector<double> cache;   // OKVIS_BA_IMU_CACHE_DOUBLES when ref_valid == 2 (kept allocated once it has been)
  };
  std::vector<Imu> imu;
  okvis_ba_imu_params imu_params{};
  // priors
  std::vector<int32_t> pprior_pose, sbprior_sb, rel_pose0, rel_pose1;
  std::vector<double> pprior_meas, pprior_sqrtinfo, sbprior_meas, sbprior_sqrtinfo, rel_sqrtinfo;
  int32_t marg_dim = 0;
  std::vector<int32_t> marg_block_type, marg_block_idx, marg_block_off;
  std::vector<double> marg_J, marg_e0, marg_lin;
  // Not part of the window's value: the flat IMU arrays view() hands out and the scratch of apply().  A copy of the container
  // starts with its own (empty) ones; a container that is copied INTO keeps its buffers, so that editing a copy
  // (okvis_ba_patch_window) allocates nothing in the steady state.
  struct Buffers {
    std::vector<int32_t> f_ip0, f_is0, f_ip1, f_is1, f_sbegin, f_scount;
    std::vector<int64_t> f_t0, f_t1, f_st;
    std::vector<double> f_gyr, f_acc, f_ref, f_cache;
    std::vector<uint8_t> f_refv;
    std::vector<int32_t> x_mp, x_ms, x_ml, x_ord, x_lm, x_pose, x_ext, x_cam;
    std::vector<double> x_uv, x_sw;
    Buffers() = default;
    Buffers(const Buffers&) {}
    Buffers& operator=(const Buffers&) { return *this; }
    Buffers(Buffers&&) = default;
    Buffers& operator=(Buffers&&) = default;
  };
  mutable Buffers buf;

  int n_pose() const { return (int)pose_fixed.size(); }
  int n_sb() const { return (int)sb_fixed.size(); }
  int n_lm() const { return (int)(lm.size() / 4); }
  int n_obs() const { return (int)obs_lm.size(); }
  int n_cam() const { return (int)cam_model.size(); }

  template <class T>
  static void put(std::vector<T>& v, const T* p, size_t n) {
    if (p && n) v.assign(p, p + n); else v.clear();
  }

  // deep copy of a caller's window (pointers are only read here)
  int assign(const okvis_ba_window& w) {
    if (w.n_pose < 0 || w.n_sb < 0 || w.n_lm < 0 || w.n_obs < 0 || w.n_imu < 0 || w.n_cam < 0 || w.n_pprior < 0 || w.n_sbprior < 0 ||
        w.n_relpose < 0 || w.marg_dim < 0 || w.n_imu_samples < 0)
      return OKVIS_BA_ERR_ARG;
    if ((w.n_pose && (!w.pose || !w.pose_fixed)) || (w.n_sb && (!w.sb || !w.sb_fixed)) || (w.n_lm && !w.lm)) return OKVIS_BA_ERR_ARG;
    if (w.n_obs && (!w.obs_lm || !w.obs_pose || !w.obs_ext || !w.obs_cam || !w.obs_uv || !w.obs_sqrtw)) return OKVIS_BA_ERR_ARG;
    if (w.n_cam && (!w.cam_intr || !w.cam_model)) return OKVIS_BA_ERR_ARG;
    if (w.n_imu && (!w.imu_pose0 || !w.imu_sb0 || !w.imu_pose1 || !w.imu_sb1 || !w.imu_t0 || !w.imu_t1 || !w.imu_s_begin ||
                    !w.imu_s_count || !w.imu_s_t || !w.imu_s_gyr || !w.imu_s_acc))
      return OKVIS_BA_ERR_ARG;
    if ((w.n_pprior && (!w.pprior_pose || !w.pprior_meas || !w.pprior_sqrtinfo)) ||
        (w.n_sbprior && (!w.sbprior_sb || !w.sbprior_meas || !w.sbprior_sqrtinfo)) ||
        (w.n_relpose && (!w.rel_pose0 || !w.rel_pose1 || !w.rel_sqrtinfo)))
      return OKVIS_BA_ERR_ARG;
    if (w.marg_dim && (!w.marg_block_type || !w.marg_block_idx || !w.marg_block_off || !w.marg_J || !w.marg_e0 || !w.marg_lin ||
                       w.marg_nblocks <= 0))
      return OKVIS_BA_ERR_ARG;
    // every index array is range-checked before anything is copied: the container is usable without a device, so
    // okvis_ba_upload's validation never sees these windows, and apply() indexes its remap tables with them
    for (int o = 0; o < w.n_obs; ++o)
      if (w.obs_lm[o] < 0 || w.obs_lm[o] >= w.n_lm || w.obs_pose[o] < 0 || w.obs_pose[o] >= w.n_pose || w.obs_ext[o] < 0 ||
          w.obs_ext[o] >= w.n_pose || w.obs_cam[o] < 0 || w.obs_cam[o] >= w.n_cam)
        return OKVIS_BA_ERR_ARG;
    for (int f = 0; f < w.n_imu; ++f) {
      if (w.imu_pose0[f] < 0 || w.imu_pose0[f] >= w.n_pose || w.imu_pose1[f] < 0 || w.imu_pose1[f] >= w.n_pose || w.imu_sb0[f] < 0 ||
          w.imu_sb0[f] >= w.n_sb || w.imu_sb1[f] < 0 || w.imu_sb1[f] >= w.n_sb)
        return OKVIS_BA_ERR_ARG;
      const int64_t b = w.imu_s_begin[f], c = w.imu_s_count[f];   // (64-bit: b + c must not wrap)
      if (b < 0 || c < 0 || b + c > (int64_t)w.n_imu_samples) return OKVIS_BA_ERR_ARG;
    }
    for (int i = 0; i < w.n_pprior; ++i)
      if (w.pprior_pose[i] < 0 || w.pprior_pose[i] >= w.n_pose) return OKVIS_BA_ERR_ARG;
    for (int i = 0; i < w.n_sbprior; ++i)
      if (w.sbprior_sb[i] < 0 || w.sbprior_sb[i] >= w.n_sb) return OKVIS_BA_ERR_ARG;
    for (int i = 0; i < w.n_relpose; ++i)
      if (w.rel_pose0[i] < 0 || w.rel_pose0[i] >= w.n_pose || w.rel_pose1[i] < 0 || w.rel_pose1[i] >= w.n_pose) return OKVIS_BA_ERR_ARG;
    if (w.marg_dim > 0)
      for (int b = 0; b < w.marg_nblocks; ++b) {
        const int t = w.marg_block_type[b];
        if (t != OKVIS_BA_BLOCK_POSE && t != OKVIS_BA_BLOCK_SPEEDBIAS) return OKVIS_BA_ERR_ARG;
        const int lim = t == OKVIS_BA_BLOCK_POSE ? w.n_pose : w.n_sb, dim = t == OKVIS_BA_BLOCK_POSE ? 6 : 9;
        if (w.marg_block_idx[b] < 0 || w.marg_block_idx[b] >= lim || w.marg_block_off[b] < 0 ||
            (int64_t)w.marg_block_off[b] + dim > (int64_t)w.marg_dim)
          return OKVIS_BA_ERR_ARG;
      }
    put(pose, w.pose, 7 * (size_t)w.n_pose); put(pose_fixed, w.pose_fixed, (size_t)w.n_pose);
    put(sb, w.sb, 9 * (size_t)w.n_sb); put(sb_fixed, w.sb_fixed, (size_t)w.n_sb);
    put(lm, w.lm, 4 * (size_t)w.n_lm);
    put(cam_intr, w.cam_intr, 12 * (size_t)w.n_cam); put(cam_model, w.cam_model, (size_t)w.n_cam);
    put(obs_lm, w.obs_lm, (size_t)w.n_obs); put(obs_pose, w.obs_pose, (size_t)w.n_obs); put(obs_ext, w.obs_ext, (size_t)w.n_obs);
    put(obs_cam, w.obs_cam, (size_t)w.n_obs); put(obs_uv, w.obs_uv, 2 * (size_t)w.n_obs); put(obs_sqrtw, w.obs_sqrtw, (size_t)w.n_obs);
    cauchy_b = w.cauchy_b;
    imu.clear();
    imu.resize((size_t)w.n_imu);
    for (int f = 0; f < w.n_imu; ++f) {
      Imu& m = imu[f];
      m.pose0 = w.imu_pose0[f]; m.sb0 = w.imu_sb0[f]; m.pose1 = w.imu_pose1[f]; m.sb1 = w.imu_sb1[f];
      m.t0 = w.imu_t0[f]; m.t1 = w.imu_t1[f];
      const int64_t b = w.imu_s_begin[f], c = w.imu_s_count[f];   // (64-bit: b + c must not wrap)
      if (b < 0 || c < 0 || b + c > (int64_t)w.n_imu_samples) return OKVIS_BA_ERR_ARG;
      put(m.s_t, w.imu_s_t + b, (size_t)c);
      put(m.s_gyr, w.imu_s_gyr + 3 * (size_t)b, 3 * (size_t)c);
      put(m.s_acc, w.imu_s_acc + 3 * (size_t)b, 3 * (size_t)c);
      m.ref_valid = (w.imu_sb_ref && w.imu_sb_ref_valid) ? w.imu_sb_ref_valid[f] : 0;
      if (m.ref_valid > 2 || (m.ref_valid == 2 && !w.imu_cache)) return OKVIS_BA_ERR_ARG;
      for (int k = 0; k < 9; ++k) m.sb_ref[k] = (m.ref_valid && w.imu_sb_ref) ? w.imu_sb_ref[9 * (size_t)f + k] : 0.0;
      if (m.ref_valid == 2) m.cache.assign(w.imu_cache + (size_t)OKVIS_BA_IMU_CACHE_DOUBLES * f, w.imu_cache + (size_t)OKVIS_BA_IMU_CACHE_DOUBLES * (f + 1));
    }
    imu_params = w.imu_params;
    put(pprior_pose, w.pprior_pose, (size_t)w.n_pprior); put(pprior_meas, w.pprior_meas, 7 * (size_t)w.n_pprior);
    put(pprior_sqrtinfo, w.pprior_sqrtinfo, 36 * (size_t)w.n_pprior);
    put(sbprior_sb, w.sbprior_sb, (size_t)w.n_sbprior); put(sbprior_meas, w.sbprior_meas, 9 * (size_t)w.n_sbprior);
    put(sbprior_sqrtinfo, w.sbprior_sqrtinfo, 81 * (size_t)w.n_sbprior);
    put(rel_pose0, w.rel_pose0, (size_t)w.n_relpose); put(rel_pose1, w.rel_pose1, (size_t)w.n_relpose);
    put(rel_sqrtinfo, w.rel_sqrtinfo, 36 * (size_t)w.n_relpose);
    set_marg(w.marg_dim, w.marg_nblocks, w.marg_block_type, w.marg_block_idx, w.marg_block_off, w.marg_J, w.marg_e0, w.marg_lin);
    return OKVIS_BA_OK;
  }
  void set_marg(int dim, int nb, const int32_t* bt, const int32_t* bi, const int32_t* bo, const double* J, const double* e0,
                const double* lin) {
    marg_dim = dim;
    if (dim <= 0) nb = 0;
    put(marg_block_type, bt, (size_t)nb); put(marg_block_idx, bi, (size_t)nb); put(marg_block_off, bo, (size_t)nb);
    put(marg_J, J, (size_t)std::max(dim, 0) * (size_t)std::max(dim, 0)); put(marg_e0, e0, (size_t)std::max(dim, 0));
    put(marg_lin, lin, 9 * (size_t)nb);
  }

  // the window as okvis_ba_upload takes it; pointers stay valid until the next assign / apply / destruction
  void view(okvis_ba_window* out) const {
    okvis_ba_window& w = *out;
    std::memset(&w, 0, sizeof(w));
    auto &f_ip0 = buf.f_ip0, &f_is0 = buf.f_is0, &f_ip1 = buf.f_ip1, &f_is1 = buf.f_is1, &f_sbegin = buf.f_sbegin, &f_scount = buf.f_scount;
    auto &f_t0 = buf.f_t0, &f_t1 = buf.f_t1, &f_st = buf.f_st;
    auto &f_gyr = buf.f_gyr, &f_acc = buf.f_acc, &f_ref = buf.f_ref;
    auto& f_refv = buf.f_refv;
    const size_t nf = imu.size();
    f_ip0.resize(nf); f_is0.resize(nf); f_ip1.resize(nf); f_is1.resize(nf); f_sbegin.resize(nf); f_scount.resize(nf);
    f_t0.resize(nf); f_t1.resize(nf); f_ref.resize(9 * nf); f_refv.resize(nf);
    auto& f_cache = buf.f_cache;
    bool any_cache = false;
    for (size_t f = 0; f < nf; ++f) any_cache = any_cache || imu[f].ref_valid == 2;
    f_cache.resize(any_cache ? (size_t)OKVIS_BA_IMU_CACHE_DOUBLES * nf : 0);
    f_st.clear(); f_gyr.clear(); f_acc.clear();
    for (size_t f = 0; f < nf; ++f) {
      const Imu& m = imu[f];
      f_ip0[f] = m.pose0; f_is0[f] = m.sb0; f_ip1[f] = m.pose1; f_is1[f] = m.sb1;
      f_t0[f] = m.t0; f_t1[f] = m.t1;
      f_sbegin[f] = (int32_t)f_st.size();
      f_scount[f] = (int32_t)m.s_t.size();
      f_st.insert(f_st.end(), m.s_t.begin(), m.s_t.end());
      f_gyr.insert(f_gyr.end(), m.s_gyr.begin(), m.s_gyr.end());
      f_acc.insert(f_acc.end(), m.s_acc.begin(), m.s_acc.end());
      std::memcpy(&f_ref[9 * f], m.sb_ref, sizeof(m.sb_ref));
      f_refv[f] = m.ref_valid;
      if (m.ref_valid == 2) std::memcpy(&f_cache[(size_t)OKVIS_BA_IMU_CACHE_DOUBLES * f], m.cache.data(), 8 * (size_t)OKVIS_BA_IMU_CACHE_DOUBLES);
    }
    w.n_pose = n_pose(); w.pose = pose.data(); w.pose_fixed = pose_fixed.data();
    w.n_sb = n_sb(); w.sb = sb.data(); w.sb_fixed = sb_fixed.data();
    w.n_lm = n_lm(); w.lm = lm.data();
    w.n_cam = n_cam(); w.cam_intr = cam_intr.data(); w.cam_model = cam_model.data();
    w.n_obs = n_obs();
    w.obs_lm = obs_lm.data(); w.obs_pose = obs_pose.data(); w.obs_ext = obs_ext.data(); w.obs_cam = obs_cam.data();
    w.obs_uv = obs_uv.data(); w.obs_sqrtw = obs_sqrtw.data();
    w.cauchy_b = cauchy_b;
    w.n_imu = (int32_t)nf;
    w.imu_pose0 = f_ip0.data(); w.imu_sb0 = f_is0.data(); w.imu_pose1 = f_ip1.data(); w.imu_sb1 = f_is1.data();
    w.imu_t0 = f_t0.data(); w.imu_t1 = f_t1.data(); w.imu_s_begin = f_sbegin.data(); w.imu_s_count = f_scount.data();
    w.n_imu_samples = (int32_t)f_st.size(); w.imu_s_t = f_st.data(); w.imu_s_gyr = f_gyr.data(); w.imu_s_acc = f_acc.data();
    w.imu_params = imu_params;
    w.imu_sb_ref = f_ref.data(); w.imu_sb_ref_valid = f_refv.data();
    w.imu_cache = any_cache ? f_cache.data() : nullptr;
    w.n_pprior = (int32_t)pprior_pose.size(); w.pprior_pose = pprior_pose.data(); w.pprior_meas = pprior_meas.data();
    w.pprior_sqrtinfo = pprior_sqrtinfo.data();
    w.n_sbprior = (int32_t)sbprior_sb.size(); w.sbprior_sb = sbprior_sb.data(); w.sbprior_meas = sbprior_meas.data();
    w.sbprior_sqrtinfo = sbprior_sqrtinfo.data();
    w.n_relpose = (int32_t)rel_pose0.size(); w.rel_pose0 = rel_pose0.data(); w.rel_pose1 = rel_pose1.data();
    w.rel_sqrtinfo = rel_sqrtinfo.data();
    w.marg_dim = marg_dim; w.marg_nblocks = (int32_t)marg_block_type.size();
    w.marg_block_type = marg_block_type.data(); w.marg_block_idx = marg_block_idx.data(); w.marg_block_off = marg_block_off.data();
    w.marg_J = marg_J.data(); w.marg_e0 = marg_e0.data(); w.marg_lin = marg_lin.data();
  }

  // ---- patch ----
  static bool ascending(const int32_t* v, int n, int limit) {
    for (int i = 0; i < n; ++i)
      if (v[i] < 0 || v[i] >= limit || (i > 0 && v[i] <= v[i - 1])) return false;
    return true;
  }
  // old index -> new index (-1 = removed); returns the number kept
  static int remap(int n, const int32_t* rem, int nrem, std::vector<int32_t>& map) {
    map.assign((size_t)n, 0);
    for (int i = 0; i < nrem; ++i) map[rem[i]] = -1;
    int k = 0;
    for (int i = 0; i < n; ++i) map[i] = map[i] < 0 ? -1 : k++;
    return k;
  }
  template <class T>
  static void compact(std::vector<T>& v, const std::vector<int32_t>& map, int width) {
    size_t o = 0;
    for (size_t i = 0; i < map.size(); ++i) {
      if (map[i] < 0) continue;
      if (o != i)
        for (int k = 0; k < width; ++k) v[o * width + k] = v[i * width + k];
      ++o;
    }
    v.resize(o * width);
  }

  int apply(const okvis_ba_patch& p) {
    const int np0 = n_pose(), ns0 = n_sb(), nl0 = n_lm(), no0 = n_obs(), ni0 = (int)imu.size();
    // ---- everything is checked before anything is changed ----
    if (p.n_remove_obs < 0 || p.n_remove_lm < 0 || p.n_remove_pose < 0 || p.n_remove_sb < 0 || p.n_remove_imu < 0 || p.n_add_pose < 0 ||
        p.n_add_sb < 0 || p.n_add_lm < 0 || p.n_add_obs < 0 || p.n_add_imu < 0 || p.n_add_imu_samples < 0 || p.n_set_pose < 0 ||
        p.n_set_sb < 0 || p.n_set_lm < 0 || p.n_pprior < 0 || p.n_sbprior < 0 || p.n_relpose < 0 || p.marg_dim < 0)
      return OKVIS_BA_ERR_ARG;
    if ((p.n_remove_obs && !p.remove_obs) || (p.n_remove_lm && !p.remove_lm) || (p.n_remove_pose && !p.remove_pose) ||
        (p.n_remove_sb && !p.remove_sb) || (p.n_remove_imu && !p.remove_imu))
      return OKVIS_BA_ERR_ARG;
    if (!ascending(p.remove_obs, p.n_remove_obs, no0) || !ascending(p.remove_lm, p.n_remove_lm, nl0) ||
        !ascending(p.remove_pose, p.n_remove_pose, np0) || !ascending(p.remove_sb, p.n_remove_sb, ns0) ||
        !ascending(p.remove_imu, p.n_remove_imu, ni0))
      return OKVIS_BA_ERR_ARG;
    if ((p.n_add_pose && (!p.add_pose || !p.add_pose_fixed)) || (p.n_add_sb && (!p.add_sb || !p.add_sb_fixed)) || (p.n_add_lm && !p.add_lm))
      return OKVIS_BA_ERR_ARG;
    if (p.n_add_obs && (!p.add_obs_lm || !p.add_obs_pose || !p.add_obs_ext || !p.add_obs_cam || !p.add_obs_uv || !p.add_obs_sqrtw))
      return OKVIS_BA_ERR_ARG;
    if (p.n_add_imu && (!p.add_imu_pose0 || !p.add_imu_sb0 || !p.add_imu_pose1 || !p.add_imu_sb1 || !p.add_imu_t0 || !p.add_imu_t1 ||
                        !p.add_imu_s_begin || !p.add_imu_s_count || !p.add_imu_s_t || !p.add_imu_s_gyr || !p.add_imu_s_acc))
      return OKVIS_BA_ERR_ARG;
    if ((p.n_set_pose && (!p.set_pose_idx || !p.set_pose)) || (p.n_set_sb && (!p.set_sb_idx || !p.set_sb)) ||
        (p.n_set_lm && (!p.set_lm_idx || !p.set_lm)))
      return OKVIS_BA_ERR_ARG;
    const bool rep_pp = (p.replace & OKVIS_BA_PATCH_POSE_PRIORS) != 0, rep_sp = (p.replace & OKVIS_BA_PATCH_SB_PRIORS) != 0,
               rep_rel = (p.replace & OKVIS_BA_PATCH_RELPOSE) != 0, rep_marg = (p.replace & OKVIS_BA_PATCH_MARG_PRIOR) != 0;
    if ((rep_pp && p.n_pprior && (!p.pprior_pose || !p.pprior_meas || !p.pprior_sqrtinfo)) ||
        (rep_sp && p.n_sbprior && (!p.sbprior_sb || !p.sbprior_meas || !p.sbprior_sqrtinfo)) ||
        (rep_rel && p.n_relpose && (!p.rel_pose0 || !p.rel_pose1 || !p.rel_sqrtinfo)))
      return OKVIS_BA_ERR_ARG;
    if (rep_marg && p.marg_dim > 0 &&
        (!p.marg_block_type || !p.marg_block_idx || !p.marg_block_off || !p.marg_J || !p.marg_e0 || !p.marg_lin || p.marg_nblocks <= 0))
      return OKVIS_BA_ERR_ARG;
    // (scratch kept between calls: an edit of a frame's worth of observations allocates nothing in the steady state)
    std::vector<int32_t>&mp = buf.x_mp, &ms = buf.x_ms, &ml = buf.x_ml;
    auto &x_ord = buf.x_ord, &x_lm = buf.x_lm, &x_pose = buf.x_pose, &x_ext = buf.x_ext, &x_cam = buf.x_cam;
    auto &x_uv = buf.x_uv, &x_sw = buf.x_sw;
    const int np1 = remap(np0, p.remove_pose, p.n_remove_pose, mp) + p.n_add_pose;
    const int ns1 = remap(ns0, p.remove_sb, p.n_remove_sb, ms) + p.n_add_sb;
    const int nlk = remap(nl0, p.remove_lm, p.n_remove_lm, ml), nl1 = nlk + p.n_add_lm;
    if (p.add_lm_before)   // places of the appended landmarks among the ones that stay: ascending, 0 .. nlk
      for (int k = 0; k < p.n_add_lm; ++k)
        if (p.add_lm_before[k] < 0 || p.add_lm_before[k] > nlk || (k > 0 && p.add_lm_before[k] < p.add_lm_before[k - 1])) return OKVIS_BA_ERR_ARG;
    if (!rep_marg)
      for (size_t b = 0; b < marg_block_type.size(); ++b)
        if ((marg_block_type[b] == OKVIS_BA_BLOCK_POSE ? mp : ms)[marg_block_idx[b]] < 0) return OKVIS_BA_ERR_ARG;
    auto in = [](int v, int n) { return v >= 0 && v < n; };
    for (int i = 0; i < p.n_add_obs; ++i)
      if (!in(p.add_obs_lm[i], nl1) || !in(p.add_obs_pose[i], np1) || !in(p.add_obs_ext[i], np1) || !in(p.add_obs_cam[i], n_cam()))
        return OKVIS_BA_ERR_ARG;
    for (int f = 0; f < p.n_add_imu; ++f) {
      if (!in(p.add_imu_pose0[f], np1) || !in(p.add_imu_pose1[f], np1) || !in(p.add_imu_sb0[f], ns1) || !in(p.add_imu_sb1[f], ns1))
        return OKVIS_BA_ERR_ARG;
      if (p.add_imu_s_begin[f] < 0 || p.add_imu_s_count[f] < 0 || p.add_imu_s_begin[f] + p.add_imu_s_count[f] > p.n_add_imu_samples)
        return OKVIS_BA_ERR_ARG;
    }
    for (int i = 0; i < p.n_set_pose; ++i) if (!in(p.set_pose_idx[i], np1)) return OKVIS_BA_ERR_ARG;
    for (int i = 0; i < p.n_set_sb; ++i) if (!in(p.set_sb_idx[i], ns1)) return OKVIS_BA_ERR_ARG;
    for (int i = 0; i < p.n_set_lm; ++i) if (!in(p.set_lm_idx[i], nl1)) return OKVIS_BA_ERR_ARG;
    if (rep_pp) for (int i = 0; i < p.n_pprior; ++i) if (!in(p.pprior_pose[i], np1)) return OKVIS_BA_ERR_ARG;
    if (rep_sp) for (int i = 0; i < p.n_sbprior; ++i) if (!in(p.sbprior_sb[i], ns1)) return OKVIS_BA_ERR_ARG;
    if (rep_rel) for (int i = 0; i < p.n_relpose; ++i) if (!in(p.rel_pose0[i], np1) || !in(p.rel_pose1[i], np1)) return OKVIS_BA_ERR_ARG;
    if (rep_marg)
      for (int b = 0; b < (p.marg_dim > 0 ? p.marg_nblocks : 0); ++b)
        if (!in(p.marg_block_idx[b], p.marg_block_type[b] == OKVIS_BA_BLOCK_POSE ? np1 : ns1)) return OKVIS_BA_ERR_ARG;

    // ---- 1 + 2: removals, stable compaction ----
    compact(pose, mp, 7); compact(pose_fixed, mp, 1);
    compact(sb, ms, 9); compact(sb_fixed, ms, 1);
    compact(lm, ml, 4);
    if (p.add_lm_before && p.n_add_lm) {
      // the appended landmarks take their places: a landmark that stays moves back by the number of new ones in front of it
      std::vector<double>& out = buf.x_uv;   // (free until the observations are merged below)
      out.resize(4 * (size_t)nl1);
      int k = 0;
      for (int j = 0; j <= nlk; ++j) {
        while (k < p.n_add_lm && p.add_lm_before[k] == j) {
          std::memcpy(&out[4 * (size_t)(j + k)], p.add_lm + 4 * (size_t)k, 32);
          ++k;
        }
        if (j < nlk) std::memcpy(&out[4 * (size_t)(j + k)], &lm[4 * (size_t)j], 32);
      }
      lm.assign(out.begin(), out.end());
      k = 0;
      int j = 0;   // (ml is ascending over the landmarks that stay)
      for (int l = 0; l < nl0; ++l) {
        if (ml[l] < 0) continue;
        while (k < p.n_add_lm && p.add_lm_before[k] <= j) ++k;
        ml[l] = j + k;
        ++j;
      }
    }
    {
      std::vector<int32_t> mi;
      remap(ni0, p.remove_imu, p.n_remove_imu, mi);
      size_t o = 0;
      for (int f = 0; f < ni0; ++f) {
        Imu& m = imu[f];
        if (mi[f] < 0 || mp[m.pose0] < 0 || mp[m.pose1] < 0 || ms[m.sb0] < 0 || ms[m.sb1] < 0) continue;
        m.pose0 = mp[m.pose0]; m.pose1 = mp[m.pose1]; m.sb0 = ms[m.sb0]; m.sb1 = ms[m.sb1];
        if (o != (size_t)f) imu[o] = std::move(m);
        ++o;
      }
      imu.resize(o);
    }
    auto keep_terms = [](std::vector<int32_t>& blk, const std::vector<int32_t>& map, std::vector<double>& a, int wa, std::vector<double>& b,
                         int wb) {
      size_t o = 0;
      for (size_t i = 0; i < blk.size(); ++i) {
        if (map[blk[i]] < 0) continue;
        blk[o] = map[blk[i]];
        if (o != i) {
          std::copy(a.begin() + i * wa, a.begin() + (i + 1) * wa, a.begin() + o * wa);
          std::copy(b.begin() + i * wb, b.begin() + (i + 1) * wb, b.begin() + o * wb);
        }
        ++o;
      }
      blk.resize(o); a.resize(o * wa); b.resize(o * wb);
    };
    if (rep_pp) {
      put(pprior_pose, p.pprior_pose, (size_t)p.n_pprior); put(pprior_meas, p.pprior_meas, 7 * (size_t)p.n_pprior);
      put(pprior_sqrtinfo, p.pprior_sqrtinfo, 36 * (size_t)p.n_pprior);
    } else {
      keep_terms(pprior_pose, mp, pprior_meas, 7, pprior_sqrtinfo, 36);
    }
    if (rep_sp) {
      put(sbprior_sb, p.sbprior_sb, (size_t)p.n_sbprior); put(sbprior_meas, p.sbprior_meas, 9 * (size_t)p.n_sbprior);
      put(sbprior_sqrtinfo, p.sbprior_sqrtinfo, 81 * (size_t)p.n_sbprior);
    } else {
      keep_terms(sbprior_sb, ms, sbprior_meas, 9, sbprior_sqrtinfo, 81);
    }
    if (rep_rel) {
      put(rel_pose0, p.rel_pose0, (size_t)p.n_relpose); put(rel_pose1, p.rel_pose1, (size_t)p.n_relpose);
      put(rel_sqrtinfo, p.rel_sqrtinfo, 36 * (size_t)p.n_relpose);
    } else {
      size_t o = 0;
      for (size_t i = 0; i < rel_pose0.size(); ++i) {
        if (mp[rel_pose0[i]] < 0 || mp[rel_pose1[i]] < 0) continue;
        rel_pose0[o] = mp[rel_pose0[i]]; rel_pose1[o] = mp[rel_pose1[i]];
        if (o != i) std::copy(rel_sqrtinfo.begin() + i * 36, rel_sqrtinfo.begin() + (i + 1) * 36, rel_sqrtinfo.begin() + o * 36);
        ++o;
      }
      rel_pose0.resize(o); rel_pose1.resize(o); rel_sqrtinfo.resize(o * 36);
    }
    if (rep_marg) {
      set_marg(p.marg_dim, p.marg_nblocks, p.marg_block_type, p.marg_block_idx, p.marg_block_off, p.marg_J, p.marg_e0, p.marg_lin);
    } else {
      for (size_t b = 0; b < marg_block_type.size(); ++b)
        marg_block_idx[b] = (marg_block_type[b] == OKVIS_BA_BLOCK_POSE ? mp : ms)[marg_block_idx[b]];
    }
    // ---- 3: appended blocks and terms ----
    pose.insert(pose.end(), p.add_pose, p.add_pose + 7 * (size_t)p.n_add_pose);
    pose_fixed.insert(pose_fixed.end(), p.add_pose_fixed, p.add_pose_fixed + p.n_add_pose);
    sb.insert(sb.end(), p.add_sb, p.add_sb + 9 * (size_t)p.n_add_sb);
    sb_fixed.insert(sb_fixed.end(), p.add_sb_fixed, p.add_sb_fixed + p.n_add_sb);
    if (!(p.add_lm_before && p.n_add_lm)) lm.insert(lm.end(), p.add_lm, p.add_lm + 4 * (size_t)p.n_add_lm);
    for (int f = 0; f < p.n_add_imu; ++f) {
      Imu m;
      m.pose0 = p.add_imu_pose0[f]; m.sb0 = p.add_imu_sb0[f]; m.pose1 = p.add_imu_pose1[f]; m.sb1 = p.add_imu_sb1[f];
      m.t0 = p.add_imu_t0[f]; m.t1 = p.add_imu_t1[f];
      const int b = p.add_imu_s_begin[f], c = p.add_imu_s_count[f];
      put(m.s_t, p.add_imu_s_t + b, (size_t)c);
      put(m.s_gyr, p.add_imu_s_gyr + 3 * (size_t)b, 3 * (size_t)c);
      put(m.s_acc, p.add_imu_s_acc + 3 * (size_t)b, 3 * (size_t)c);
      m.ref_valid = 0;   // a brand-new term: its first evaluation re-preintegrates at the current bias (ImuError.cpp:62)
      std::memset(m.sb_ref, 0, sizeof(m.sb_ref));
      imu.push_back(std::move(m));
    }
    // ---- 2 + 4: observations in ONE pass: what stays is renumbered (its order is unchanged by the compaction of the blocks),
    //      what comes is merged into the (landmark, pose, camera) order behind equal keys ----
    if (p.n_add_obs || (size_t)no0 != 0) {
      std::vector<int32_t>& ord = x_ord;
      ord.resize((size_t)p.n_add_obs);
      for (int i = 0; i < p.n_add_obs; ++i) ord[i] = i;
      auto key_less = [](int l0, int p0, int c0, int l1, int p1, int c1) {
        if (l0 != l1) return l0 < l1;
        if (p0 != p1) return p0 < p1;
        return c0 < c1;
      };
      auto new_less = [&](int a, int b) {
        return key_less(p.add_obs_lm[a], p.add_obs_pose[a], p.add_obs_cam[a], p.add_obs_lm[b], p.add_obs_pose[b], p.add_obs_cam[b]);
      };
      if (!std::is_sorted(ord.begin(), ord.end(), new_less)) std::stable_sort(ord.begin(), ord.end(), new_less);
      const size_t cap = (size_t)no0 + (size_t)p.n_add_obs;
      x_lm.resize(cap); x_pose.resize(cap); x_ext.resize(cap); x_cam.resize(cap); x_uv.resize(2 * cap); x_sw.resize(cap);
      // (plain pointers: the loop below is the bulk of an edit's time — a frame's worth of records moved once)
      int32_t* __restrict o_lm = x_lm.data(); int32_t* __restrict o_pose = x_pose.data(); int32_t* __restrict o_ext = x_ext.data();
      int32_t* __restrict o_cam = x_cam.data(); double* __restrict o_uv = x_uv.data(); double* __restrict o_sw = x_sw.data();
      const int32_t* __restrict i_lm = obs_lm.data(); const int32_t* __restrict i_pose = obs_pose.data();
      const int32_t* __restrict i_ext = obs_ext.data(); const int32_t* __restrict i_cam = obs_cam.data();
      const double* __restrict i_uv = obs_uv.data(); const double* __restrict i_sw = obs_sqrtw.data();
      const int32_t* __restrict r_ml = ml.data(); const int32_t* __restrict r_mp = mp.data();
      const int32_t* __restrict rem = p.remove_obs;   // (ascending: a cursor instead of a map over all observations)
      int ro = 0;
      const int nrem = p.n_remove_obs;
      size_t o = 0;
      int j = 0;
      const int nadd = p.n_add_obs;
      auto put_new = [&](int a) {
        o_lm[o] = p.add_obs_lm[a]; o_pose[o] = p.add_obs_pose[a]; o_ext[o] = p.add_obs_ext[a]; o_cam[o] = p.add_obs_cam[a];
        o_uv[2 * o] = p.add_obs_uv[2 * (size_t)a]; o_uv[2 * o + 1] = p.add_obs_uv[2 * (size_t)a + 1]; o_sw[o] = p.add_obs_sqrtw[a];
        ++o;
      };
      // key of the next record that comes (INT_MAX landmark once there is none: nothing is in front of it any more)
      int nl = 0x7fffffff, np_ = 0, nc = 0;
      auto next_key = [&] {
        if (j < nadd) nl = p.add_obs_lm[ord[j]], np_ = p.add_obs_pose[ord[j]], nc = p.add_obs_cam[ord[j]];
        else nl = 0x7fffffff;
      };
      next_key();
      for (int i = 0; i < no0; ++i) {
        if (ro < nrem && rem[ro] == i) {
          ++ro;
          continue;
        }
        const int l = r_ml[i_lm[i]], ip = r_mp[i_pose[i]], ie = r_mp[i_ext[i]], c = i_cam[i];
        if ((l | ip | ie) < 0) continue;   // observations of a removed landmark / pose / extrinsics block go with it
        while (nl < l || (nl == l && (np_ < ip || (np_ == ip && nc < c)))) {
          put_new(ord[j++]);
          next_key();
        }
        o_lm[o] = l; o_pose[o] = ip; o_ext[o] = ie; o_cam[o] = c;
        o_uv[2 * o] = i_uv[2 * (size_t)i]; o_uv[2 * o + 1] = i_uv[2 * (size_t)i + 1]; o_sw[o] = i_sw[i];
        ++o;
      }
      while (j < nadd) put_new(ord[j++]);
      x_lm.resize(o); x_pose.resize(o); x_ext.resize(o); x_cam.resize(o); x_uv.resize(2 * o); x_sw.resize(o);
      obs_lm.swap(x_lm); obs_pose.swap(x_pose); obs_ext.swap(x_ext); obs_cam.swap(x_cam); obs_uv.swap(x_uv); obs_sqrtw.swap(x_sw);
    }
    // ---- 5: sparse values ----
    for (int i = 0; i < p.n_set_pose; ++i) std::memcpy(&pose[7 * (size_t)p.set_pose_idx[i]], p.set_pose + 7 * (size_t)i, 56);
    for (int i = 0; i < p.n_set_sb; ++i) std::memcpy(&sb[9 * (size_t)p.set_sb_idx[i]], p.set_sb + 9 * (size_t)i, 72);
    for (int i = 0; i < p.n_set_lm; ++i) std::memcpy(&lm[4 * (size_t)p.set_lm_idx[i]], p.set_lm + 4 * (size_t)i, 32);
    return OKVIS_BA_OK;
  }

  // values of an optimisation written back (packed record of okvis_ba_fetch_results: pose | sb | lm | quality | IMU reference biases)
  // take_refs: the device has evaluated the IMU terms since the upload (their caches carry the bias they were built at)
  void take_results(const unsigned char* rec, bool take_refs) {
    const size_t b_pose = 56 * (size_t)n_pose(), b_sb = 72 * (size_t)n_sb(), b_lm = 32 * (size_t)n_lm(), b_q = 8 * (size_t)n_lm();
    if (b_pose) std::memcpy(pose.data(), rec, b_pose);
    if (b_sb) std::memcpy(sb.data(), rec + b_pose, b_sb);
    if (b_lm) std::memcpy(lm.data(), rec + b_pose + b_sb, b_lm);
    const unsigned char* r = rec + b_pose + b_sb + b_lm + b_q;
    const unsigned char* rc = r + 72 * imu.size();   // the preintegration records behind the reference biases
    for (size_t f = 0; take_refs && f < imu.size(); ++f) {
      std::memcpy(imu[f].sb_ref, r + 72 * f, 72);
      imu[f].ref_valid = 1;
      const unsigned char* c = rc + 8 * (size_t)OKVIS_BA_IMU_CACHE_DOUBLES * f;
      int32_t valid;
      std::memcpy(&valid, c + 8 * (size_t)(OKVIS_BA_IMU_CACHE_DOUBLES - 1), 4);   // (the record's last double holds its two flag words)
      if (valid == 1) {
        imu[f].cache.assign(reinterpret_cast<const double*>(c), reinterpret_cast<const double*>(c) + OKVIS_BA_IMU_CACHE_DOUBLES);
        imu[f].ref_valid = 2;
      }
    }
  }
};

}  // namespace ba
