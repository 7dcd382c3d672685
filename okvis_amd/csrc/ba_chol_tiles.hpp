// Dense Cholesky solve of the reduced camera system for windows too large for the single-workgroup LDS
// solver (D > MAX_D_LDS, e.g. BASELINE configs[2]: 50 keyframes, D = 750): a tiled, multi-workgroup,
// left-looking factorisation whose tile updates are fp64 MFMA GEMMs (v_mfma_f64_16x16x4_f64).
//
//   tiles      48 x 48 (8 blocks of the 6-wide pose/speed-bias granularity, 3 MFMA sub-tiles of 16), row-major,
//              lower triangle only, tile (i,j) at ((i(i+1)/2 + j) * 48 * 48
//   one workgroup per tile (the tile left of a diagonal tile is made by the diagonal tile's workgroup, see chol_tile_task),
//   launched in COLUMN-major task order.  Tile (i,j) needs tiles (i,k), (j,k), k < j,
//   and the diagonal tile (j,j): all of them have a smaller task index.  Workgroups are dispatched in index
//   order, so every dependency is resident or finished when a workgroup starts spinning on its flag: no
//   host synchronisation, no cooperative launch, no deadlock.  (Spins are bounded anyway: a stuck
//   dependency marks the factorisation as failed instead of hanging the GPU.)
//
//     off-diagonal (i,j):  C = A_ij - sum_k L_ik L_jk^T  (MFMA);  L_ij = C L_jj^-T = C (Linv_j)^T  (MFMA)
//     diagonal (j,j):      C = A_jj - sum_k L_jk L_jk^T  (MFMA);  L_jj^-1 in LDS (ct_ldl_inv48),
//                          Linv_j = L_jj^-1 (published for the column's TRSMs and for the back-substitution),
//                          y_j = Linv_j (rhs_j - sum_k L_jk y_k)  (forward substitution rides along)
//   back-substitution L^T x = y: nT more tasks behind the tiles (chol_backsub_task): workgroup b owns tile row j = nT-1-b,
//   keeps t_j = y_j - sum_(i>j) L_ij^T x_i, takes every x_i as it appears (the VALUES are polled: x is pre-set to a sentinel,
//   so a value and its readiness are one memory round trip), holds Linv_j and the last tile it will need, L_(j+1,j), in LDS
//   and prefetches the others - the serial chain is one poll + two 48x48 products from LDS per tile row.
#pragma once
#include "ba_device.hpp"
#include "ba_ldl16.hpp"
#include "ba_types.hpp"

namespace ba {

constexpr int CT_TB = 48;            // tile edge
constexpr int CT_LD = 49;            // LDS row stride (doubles): conflict-free column walks
constexpr int CT_THREADS = 256;
constexpr int CT_TILE = CT_TB * CT_TB;
constexpr int CT_TILE_BYTES = CT_TILE * 8;
constexpr int CT_SPIN_LIMIT = 1 << 22;

typedef double ct_v4 __attribute__((ext_vector_type(4)));

struct CholTiles {   // per-window workspace of the tiled solver (device pointers)
  int nT;            // tile rows/columns; padded dimension = 48 nT
  double* T;         // lower tiles, A on entry, L on exit
  double* Linv;      // [nT] inverse of the diagonal tiles
  double* rhs;       // [48 nT] right-hand side on entry
  double* y;         // [48 nT] L^-1 rhs
  int* flag;         // [nT(nT+1)/2 + 1]: tile done flags; last entry = failure (non-PD pivot / dependency timeout)
  int* pflag = nullptr;      // [2 nT] "partial ready" flags of the tasks that prepare the chain's inputs: [j] tile (j,j), [nT + j] tile (j+1,j)
  int* progress = nullptr;   // number of diagonal tiles published so far (null: every wait polls its flag from the start)
  double* x = nullptr;   // [48 nT] solution; when set, tasks nT(nT+1)/2 .. + nT - 1 are the back-substitution (below)
  double* tl = nullptr;  // diagnostics: per task 4 wall_clock64() stamps (start, dependencies met, own work done, flag set)
};

__device__ __forceinline__ int ct_tile_index(int i, int j) { return i * (i + 1) / 2 + j; }

// Everything one workgroup hands to another inside this launch travels through DEVICE-COHERENT accesses (relaxed agent-scope
// atomics: written through / read past the XCD's L2) and a flag raised after the writer's stores have been performed
// (ct_release = s_waitcnt).  The textbook pair — plain stores, agent-scope release, acquire on the reader — writes the writer's
// whole L2 back and invalidates the reader's at every poll: 3.5 us per hand-over when the chip is quiet, 10 us in the first
// steps of the factorisation when 130 workgroups poll (profiles/r03_notes.md).
__device__ __forceinline__ void ct_gst(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ct_gld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// drain: every wave that stored a payload waits for its stores before the workgroup's barrier and the flag behind it
__device__ __forceinline__ void ct_release() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// 16-byte device-coherent accesses through a buffer descriptor (aux 16 = sc1); `base` must be wave-uniform
typedef unsigned int ct_v4u __attribute__((ext_vector_type(4)));
typedef double ct_v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ct_v2 ct_gld2(const double* base, int pair) {
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, CT_TILE_BYTES, 0x00020000);
  return __builtin_bit_cast(ct_v2, __builtin_amdgcn_raw_buffer_load_b128(rs, pair * 16, 0, 16));
}
__device__ __forceinline__ void ct_gst2(double* base, int pair, double a, double b) {
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, CT_TILE_BYTES, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ct_v4u, ct_v2{a, b}), rs, pair * 16, 0, 16);
}
__device__ __forceinline__ void ct_raise(int* f, int v = 1) { __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// wait until *f != 0 (acquire); returns false on timeout
__device__ __forceinline__ bool ct_wait(const int* f) {
  for (int it = 0; it < CT_SPIN_LIMIT; ++it) {
    if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;
    __builtin_amdgcn_s_sleep(2);
  }
  return false;
}
// the same for a tile of column k (it cannot exist before diagonal tile k - 1 is published): far from its turn the waiter looks
// at the progress counter every microsecond or so instead of hammering the flag — all ~150 workgroups of a window are resident
// from the start, and their polls share a handful of cache lines (measured: the first steps of the diagonal chain took 35 us,
// the last ones 17.5, for the same work)
__device__ __forceinline__ bool ct_wait_col(const CholTiles& C, const int* f, int k) {
  if (C.progress) {
    int it = 0;
    while (__hip_atomic_load(C.progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k && it < CT_SPIN_LIMIT) {
      __builtin_amdgcn_s_sleep(100);
      ++it;
    }
  }
  return ct_wait(f);
}

// global tile (row-major 48x48) -> LDS (stride CT_LD)
__device__ __forceinline__ void ct_load_tile(const double* g, double* l, int tid) {
  for (int c = tid; c < CT_TILE / 2; c += CT_THREADS) {   // (48 is even: a pair never straddles two rows)
    const ct_v2 v = ct_gld2(g, c);
    const int e = 2 * c;
    double* d = l + (e / CT_TB) * CT_LD + (e % CT_TB);
    d[0] = v[0];
    d[1] = v[1];
  }
}

// acc(strip r of 16 rows, 3 sub-tiles of 16 columns) += sign * A[16r.., :] * B^T   with A, B 48x48 in LDS.
// v_mfma_f64_16x16x4_f64: A operand lane l = A[row l&15][k l>>4], B operand lane l = B[k l>>4][col l&15],
// C/D lane l, register i = C[row (l>>4) + 4 i][col l&15]  (cdna_hip_programming.md "Fragment layout")
__device__ __forceinline__ void ct_gemm_nt(ct_v4 acc[3], const double* A, const double* B, int strip, int lane, double sign) {
  const int rr = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int k0 = 0; k0 < CT_TB; k0 += 4) {
    const double a = sign * A[(16 * strip + rr) * CT_LD + k0 + kk];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double b = B[(16 * c + rr) * CT_LD + k0 + kk];   // B^T[k][col] = B[col][k]
      acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
  }
}

// accumulator strip -> LDS / global (row-major with leading dimension ld)
__device__ __forceinline__ void ct_store_acc(const ct_v4 acc[3], double* dst, int ld, int strip, int lane) {
  const int col = lane & 15, r0 = lane >> 4;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[(16 * strip + r0 + 4 * i) * ld + 16 * c + col] = acc[c][i];
}
__device__ __forceinline__ void ct_load_acc(ct_v4 acc[3], const double* src, int ld, int strip, int lane) {
  const int col = lane & 15, r0 = lane >> 4;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[c][i] = src[(16 * strip + r0 + 4 * i) * ld + 16 * c + col];
}

// the same from / to global memory (device-coherent, see ct_gst)
__device__ __forceinline__ void ct_store_acc_g(const ct_v4 acc[3], double* dst, int strip, int lane) {
  const int col = lane & 15, r0 = lane >> 4;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) ct_gst(dst + (16 * strip + r0 + 4 * i) * CT_TB + 16 * c + col, acc[c][i]);
}
__device__ __forceinline__ void ct_load_acc_g(ct_v4 acc[3], const double* src, int strip, int lane) {
  const int col = lane & 15, r0 = lane >> 4;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[c][i] = ct_gld(src + (16 * strip + r0 + 4 * i) * CT_TB + 16 * c + col);
}

__device__ __forceinline__ double ct_rsqrt(double x) {  // v_rsq_f64 + two Newton steps: full fp64 accuracy
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
}

// The inverse Cholesky factor of a diagonal tile through a blocked LDL^T with 16-wide panels: the three 16x16 diagonal
// blocks are eliminated by ONE wave in registers (ldl16_eliminate, ba_ldl16.hpp: column per lane, pivot column by DPP
// row_newbcast, the unit-lower inverse in the same registers), everything between them is 16x16x16 products by all 256
// work-items from LDS.  Twelve barrier phases of ~0.3 us and three eliminations of ~0.85 us: 9 us per tile (rounds 1-2: a
// right-looking factorisation with 6-wide block columns, the diagonal block factored and inverted by one work-item, 20.5 us).
//   M: 48x48 SPD in LDS (stride CT_LD, lower triangle referenced; overwritten).  X: L^-1 with A = L L^T (full square, zeros
//   above the diagonal) = D^-1/2 Lt^-1 for A = Lt D Lt^T.  R: 48 x CT_LD scratch.  dinv: 48 doubles scratch.
__device__ void ct_ldl_inv48(double* M, double* X, double* R, double* dinv, int tid, int* s_fail) {
  const int lane = tid & 63, wave = tid >> 6, j = lane & 15;
  // ---- one elimination step of the chain: diagonal block b (rows / columns 16 b ..) of M -> unit-lower inverse into X
  auto eliminate = [&](int b) {
    if (wave == 0) {
      double c[16];
      double mine = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {   // full symmetric block from the lower triangle
        const int hi = i > j ? i : j, lo = i > j ? j : i;
        c[i] = M[(16 * b + hi) * CT_LD + 16 * b + lo];
      }
      bool bad = false;
      // (guarded: a pivot that is not positive becomes 1, the tile reports failure.  Not compensated — ba_ldl16.hpp, ldl16_pivot<.., COMP>:
      //  a second instantiation in this kernel costs 41 registers, scratch and half the occupancy, configs[2] 327 -> 398 us per solve,
      //  and two D = 300 windows showed nothing to gain against the long double referee, profiles/r05_referee_large.txt)
      ldl16_eliminate<true, true>(c, 16, mine, j, &bad);
      if (bad && lane == 0) *s_fail = 1;
      if (lane < 16) {
        dinv[16 * b + j] = mine;
#pragma unroll
        for (int i = 0; i < 16; ++i) X[(16 * b + i) * CT_LD + 16 * b + j] = i > j ? -c[i] * mine : (i == j ? 1.0 : 0.0);   // (c: -L^-1 D below the diagonal)
      }
    }
  };
  // one 16x16x16 product per wave on the matrix core: C(i, n) = sum_k A(i, k) B(k, n), A / B / C given by element functions
  // (lane l supplies A(l & 15, (l >> 4) + 4 q) and B((l >> 4) + 4 q, l & 15), receives C((l >> 4) + 4 r, l & 15))
  auto product = [&](int w, auto&& a_at, auto&& b_at, auto&& store) {
    if (wave != w) return;
    double a[4], bb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a[q] = a_at(lane & 15, (lane >> 4) + 4 * q);
      bb[q] = b_at((lane >> 4) + 4 * q, lane & 15);
    }
    ct_v4 acc{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], bb[q], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) store((lane >> 4) + 4 * r, lane & 15, acc[r]);
  };
  eliminate(0);
  __syncthreads();
  // R_0J = Lt_0^-1 A_0J (J = 1, 2), A_0J(k, n) = M(16 J + n, k)
  for (int J = 1; J <= 2; ++J)
    product(J, [&](int i, int k) { return X[i * CT_LD + k]; }, [&](int k, int n) { return M[(16 * J + n) * CT_LD + k]; },
            [&](int i, int n, double v) { R[i * CT_LD + 16 * J + n] = v; });
  __syncthreads();
  // trailing update of rows / columns 16..47 (lower blocks (1,1), (2,1), (2,2)):  A_IJ -= R_0I^T D_0^-1 R_0J
  {
    const int I = wave == 0 ? 1 : 2, J = wave <= 1 ? 1 : 2;   // wave 0: (1,1), wave 1: (2,1), wave 2: (2,2)
    product(wave < 3 ? wave : -1, [&](int i, int k) { return R[k * CT_LD + 16 * I + i] * dinv[k]; }, [&](int k, int n) { return R[k * CT_LD + 16 * J + n]; },
            [&](int i, int n, double v) { if (I > J || n <= i) M[(16 * I + i) * CT_LD + 16 * J + n] -= v; });
  }
  __syncthreads();
  eliminate(1);
  __syncthreads();
  product(1, [&](int i, int k) { return X[(16 + i) * CT_LD + 16 + k]; }, [&](int k, int n) { return M[(32 + n) * CT_LD + 16 + k]; },
          [&](int i, int n, double v) { R[(16 + i) * CT_LD + 32 + n] = v; });
  __syncthreads();
  product(0, [&](int i, int k) { return R[(16 + k) * CT_LD + 32 + i] * dinv[16 + k]; }, [&](int k, int n) { return R[(16 + k) * CT_LD + 32 + n]; },
          [&](int i, int n, double v) { if (n <= i) M[(32 + i) * CT_LD + 32 + n] -= v; });
  __syncthreads();
  eliminate(2);
  __syncthreads();
  // ---- the off-diagonal blocks of Lt^-1 (Y): with Lt_IJ = R_JI^T D_J^-1,
  //   Y10 = -Y11 (Lt10 Y00),  Y21 = -Y22 (Lt21 Y11),  Y20 = -Y22 (Lt20 Y00 + Lt21 Y10)
  // T blocks go to the (dead) lower-left part of R: T10 at rows 16.., T21 at rows 32.. columns 16.., T20 at rows 32.. columns 0..
  product(1, [&](int i, int k) { return R[k * CT_LD + 16 + i] * dinv[k]; }, [&](int k, int n) { return X[k * CT_LD + n]; },
          [&](int i, int n, double v) { R[(16 + i) * CT_LD + n] = v; });
  product(2, [&](int i, int k) { return R[(16 + k) * CT_LD + 32 + i] * dinv[16 + k]; }, [&](int k, int n) { return X[(16 + k) * CT_LD + 16 + n]; },
          [&](int i, int n, double v) { R[(32 + i) * CT_LD + 16 + n] = v; });
  product(3, [&](int i, int k) { return R[k * CT_LD + 32 + i] * dinv[k]; }, [&](int k, int n) { return X[k * CT_LD + n]; },
          [&](int i, int n, double v) { R[(32 + i) * CT_LD + n] = v; });
  __syncthreads();
  product(1, [&](int i, int k) { return X[(16 + i) * CT_LD + 16 + k]; }, [&](int k, int n) { return R[(16 + k) * CT_LD + n]; },
          [&](int i, int n, double v) { X[(16 + i) * CT_LD + n] = -v; });
  product(2, [&](int i, int k) { return X[(32 + i) * CT_LD + 32 + k]; }, [&](int k, int n) { return R[(32 + k) * CT_LD + 16 + n]; },
          [&](int i, int n, double v) { X[(32 + i) * CT_LD + 16 + n] = -v; });
  __syncthreads();
  // T20 += Lt21 Y10
  product(3, [&](int i, int k) { return R[(16 + k) * CT_LD + 32 + i] * dinv[16 + k]; }, [&](int k, int n) { return X[(16 + k) * CT_LD + n]; },
          [&](int i, int n, double v) { R[(32 + i) * CT_LD + n] += v; });
  __syncthreads();
  product(3, [&](int i, int k) { return X[(32 + i) * CT_LD + 32 + k]; }, [&](int k, int n) { return R[(32 + k) * CT_LD + n]; },
          [&](int i, int n, double v) { X[(32 + i) * CT_LD + n] = -v; });
  __syncthreads();
  // L^-1 = D^-1/2 Lt^-1: rows scaled, zeros above the diagonal blocks
  for (int e = tid; e < CT_TB * CT_TB; e += CT_THREADS) {
    const int r = e / CT_TB, c = e - r * CT_TB;
    X[r * CT_LD + c] = (c >> 4) > (r >> 4) ? 0.0 : X[r * CT_LD + c] * sqrt(dinv[r]);
  }
  __syncthreads();
}

// x is pre-set to this pattern (a signalling-NaN payload no computation produces); a slot that holds anything else is final
constexpr unsigned long long CT_X_SENTINEL = 0x7ff4dead0badf00dULL;

// back-substitution task for tile row j (see the header comment).  256 work-items; lds: three 48 x CT_LD tiles.
__device__ void chol_backsub_task(const CholTiles& C, int j, double* lds) {
  const int tid = threadIdx.x, nT = C.nT;
  double* sL = lds;                         // Linv_j
  double* sNear = lds + CT_TB * CT_LD;      // tile (j+1, j): the last one this row needs, i.e. the one on the serial chain
  double* sFar = lds + 2 * CT_TB * CT_LD;   // tiles (i, j), i > j+1, one at a time
  __shared__ double s_t[CT_TB], s_xi[CT_TB], s_part[5][CT_TB];
  __shared__ int s_ok;
  int* failflag = C.flag + nT * (nT + 1) / 2;
  const unsigned long long* xs = reinterpret_cast<const unsigned long long*>(C.x);
  auto poll = [&](int slot) {
    unsigned long long v;
    int it = 0;
    do {
      v = __atomic_load_n(xs + slot, __ATOMIC_RELAXED);
      if (v != CT_X_SENTINEL) break;
      __builtin_amdgcn_s_sleep(1);
    } while (++it < CT_SPIN_LIMIT);
    return v;
  };
  if (tid == 0) {
    bool ok = ct_wait_col(C, C.flag + ct_tile_index(j, j), j);
    if (ok && j + 1 < nT) ok = ct_wait(C.flag + ct_tile_index(j + 1, j));
    s_ok = ok ? 1 : 0;
  }
  __syncthreads();
  if (!s_ok) {
    if (tid == 0) ct_raise(failflag);
    for (int c = tid; c < CT_TB; c += CT_THREADS) C.x[CT_TB * j + c] = 0.0;   // let the rows above run out
    return;
  }
  ct_load_tile(C.Linv + (size_t)j * CT_TILE, sL, tid);
  if (j + 1 < nT) ct_load_tile(C.T + (size_t)ct_tile_index(j + 1, j) * CT_TILE, sNear, tid);
  if (tid < CT_TB) s_t[tid] = ct_gld(C.y + CT_TB * j + tid);
  constexpr int PER = CT_TILE / CT_THREADS;   // 9 entries of a tile per work-item
  double pre[PER];
  auto fetch = [&](int i) {
    const double* g = C.T + (size_t)ct_tile_index(i, j) * CT_TILE;
#pragma unroll
    for (int k = 0; k < PER; ++k) pre[k] = ct_gld(g + tid + k * CT_THREADS);
  };
  const int c = tid % CT_TB, p = tid / CT_TB;   // (component of a 48-vector, part 0..4 of the rows; p = 5: idle)
  if (nT - 1 >= j + 2) {
    // the far tiles of the column are final at the latest when the factorisation is, i.e. when x_(nT-1) appears
    if (tid == 0) (void)poll(CT_TB * (nT - 1));
    __syncthreads();
    fetch(nT - 1);   // (device-coherent loads: the tiles were performed before their flags, the flags before x_(nT-1))
  }
  for (int i = nT - 1; i > j; --i) {
    const double* cur = (i == j + 1) ? sNear : sFar;
    if (i != j + 1) {   // park the fetched tile (the previous step's reads of sFar ended before its second barrier)
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int e = tid + k * CT_THREADS;
        sFar[(e / CT_TB) * CT_LD + (e % CT_TB)] = pre[k];
      }
    }
    if (i - 1 >= j + 2) fetch(i - 1);   // next far tile: in flight while this step waits and computes
    if (tid < CT_TB) s_xi[tid] = __longlong_as_double((long long)poll(CT_TB * i + tid));   // x_i, value-polled
    __syncthreads();
    if (p < 5) {                         // t_j -= L_ij^T x_i
      double a = 0;
      for (int r = p; r < CT_TB; r += 5) a += cur[r * CT_LD + c] * s_xi[r];
      s_part[p][c] = a;
    }
    __syncthreads();
    if (tid < CT_TB) s_t[tid] -= s_part[0][tid] + s_part[1][tid] + s_part[2][tid] + s_part[3][tid] + s_part[4][tid];
  }
  __syncthreads();
  if (p < 5) {             // x_j = Linv_j^T t_j
    double a = 0;
    for (int r = p; r < CT_TB; r += 5)
      if (r >= c) a += sL[r * CT_LD + c] * s_t[r];
    s_part[p][c] = a;
  }
  __syncthreads();
  if (tid < CT_TB) {
    double v = s_part[0][tid] + s_part[1][tid] + s_part[2][tid] + s_part[3][tid] + s_part[4][tid];
    if (__double_as_longlong(v) == (long long)CT_X_SENTINEL) v = __longlong_as_double(0x7ff8000000000000LL);   // a NaN stays a NaN
    // (x_j is all a reader of x_j needs: no fence)
    if (C.tl && tid == 0) C.tl[4 * (nT * (nT + 1) / 2 + nT - 1 - j) + 3] = (double)wall_clock64();
    __atomic_store_n(reinterpret_cast<unsigned long long*>(C.x) + CT_TB * j + tid, (unsigned long long)__double_as_longlong(v),
                     __ATOMIC_RELAXED);
  }
}

// ---- the diagonal chain.  ONE workgroup (task 0) walks all diagonal tiles: factor + inverse of tile j in LDS, L_(j+1,j) = C2
// Linv_j^T and the last update of tile j+1 straight from LDS — the hand-over of Linv_j and L_(j+1,j) between two workgroups
// through memory is off the serial chain.  What it consumes is prepared by the tasks whose grid slots used to do that work:
//   task (j,j), j >= 1   publishes  A_jj - sum_(k<j-1) L_jk L_jk^T  (in place) and  rhs_j - sum_(k<j-1) L_jk y_k  (in place)
//   task (j+1,j)         publishes  A_(j+1,j) - sum_(k<j) L_(j+1,k) L_jk^T  (in place)
// with "partial ready" flags (C.pflag); the chain overwrites them with L_jj / L_(j+1,j) and raises the ordinary flags.
__device__ void chol_chain_task(const CholTiles& C, double* lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nT = C.nT;
  double* sA = lds;                       // scratch of the factorisation, then C2_(j+1,j)
  double* sB = lds + CT_TB * CT_LD;       // Linv_j
  double* sC = lds + 2 * CT_TB * CT_LD;   // tile j (L_jj after the factorisation)
  double* sL = lds + 3 * CT_TB * CT_LD;   // L_(j+1,j)
  __shared__ int s_ok, s_fail, s_c2;
  __shared__ double s_r[CT_TB], s_y[CT_TB], s_dinv[CT_TB];
  int* failflag = C.flag + nT * (nT + 1) / 2;
  if (tid == 0) {
    s_ok = 1;
    s_fail = 0;
    s_c2 = 0;
  }
  ct_v4 acc[3];
  if (wave < 3) ct_load_acc_g(acc, C.T, wave, lane);
  if (tid < CT_TB) s_r[tid] = C.rhs[tid];
  __syncthreads();
  for (int j = 0; j < nT; ++j) {
    const int tjj = ct_tile_index(j, j);
    int task_jj = 0;
    for (int c = 0; c < j; ++c) task_jj += nT - c;
    if (C.tl && tid == 0) C.tl[4 * task_jj + 1] = (double)wall_clock64();
    // ---- factor + inverse of tile j
    if (wave < 3) ct_store_acc(acc, sC, CT_LD, wave, lane);
    ct_release();   // (the stores of L_(j,j-1), issued two phases ago, have been performed)
    __syncthreads();
    if (tid == 0 && j > 0) ct_raise(C.flag + ct_tile_index(j, j - 1));
    ct_ldl_inv48(sC, sB, sA, s_dinv, tid, &s_fail);
    if (C.tl && tid == 0) C.tl[4 * task_jj + 2] = (double)wall_clock64();
    if (tid < CT_TB) {   // y_j = Linv_j r_j
      double v = 0;
      for (int m = 0; m <= tid; ++m) v += sB[tid * CT_LD + m] * s_r[m];
      s_y[tid] = v;
    }
    __syncthreads();
    const bool more = j + 1 < nT;
    // ---- wave 3 sends Linv_j and y_j off (write-through stores + drain) while the others wait for / fetch C2_(j+1,j)
    double* Tsub = C.T + (size_t)ct_tile_index(more ? j + 1 : j, j) * CT_TILE;
    if (wave == 3) {
      double* Linv = C.Linv + (size_t)j * CT_TILE;
      for (int c = lane; c < CT_TILE / 2; c += 64) {
        const int e = 2 * c;
        const double* q = sB + (e / CT_TB) * CT_LD + (e % CT_TB);
        ct_gst2(Linv, c, q[0], q[1]);
      }
      if (lane < CT_TB) ct_gst(C.y + CT_TB * j + lane, s_y[lane]);
      ct_release();
      if (lane == 0) {
        if (s_fail) ct_raise(failflag);
        ct_raise(C.flag + tjj);
        if (C.progress) ct_raise(C.progress, j + 1);
        if (C.tl) C.tl[4 * task_jj + 3] = (double)wall_clock64();
      }
    } else if (more) {
      // waves 0 - 2 fetch C2_(j+1,j) meanwhile; they follow work-item 0's poll through an LDS word, not through a barrier (which
      // would wait for the publishing wave)
      if (tid == 0) {
        const bool ok = ct_wait(C.pflag + nT + j);   // C2_(j+1,j) ready (long ago, normally)
        if (!ok) s_ok = 0;
        __hip_atomic_store(&s_c2, j + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      while (__hip_atomic_load(&s_c2, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != j + 1) __builtin_amdgcn_s_sleep(1);
      if (s_ok)
        for (int c = tid; c < CT_TILE / 2; c += 192) {
          const ct_v2 v = ct_gld2(Tsub, c);
          const int e = 2 * c;
          double* d = sA + (e / CT_TB) * CT_LD + (e % CT_TB);
          d[0] = v[0];
          d[1] = v[1];
        }
    }
    if (!more) break;
    __syncthreads();
    if (!s_ok) break;
    // ---- L_(j+1,j) = C2 Linv_j^T: to LDS for the update below, to memory for column j's tiles and the back-substitution
    //      (its flag is raised at the top of the next round, when the stores have long been performed)
    if (wave < 3) {
      ct_v4 l2[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) l2[c] = ct_v4{0.0, 0.0, 0.0, 0.0};
      ct_gemm_nt(l2, sA, sB, wave, lane, 1.0);
      ct_store_acc(l2, sL, CT_LD, wave, lane);
      ct_store_acc_g(l2, Tsub, wave, lane);
    } else if (tid == 192) {
      if (!ct_wait(C.pflag + j + 1)) s_ok = 0;   // partial of tile j+1 ready
    }
    __syncthreads();
    if (!s_ok) break;
    // ---- tile j+1 = its published partial - L L^T;  r_(j+1) = its published partial - L y_j
    if (wave < 3) {
      ct_load_acc_g(acc, C.T + (size_t)ct_tile_index(j + 1, j + 1) * CT_TILE, wave, lane);
      ct_gemm_nt(acc, sL, sL, wave, lane, -1.0);
    } else if (lane < CT_TB) {
      double v = ct_gld(C.rhs + CT_TB * (j + 1) + lane);
      for (int m = 0; m < CT_TB; ++m) v -= sL[lane * CT_LD + m] * s_y[m];
      s_r[lane] = v;
    }
    // (no barrier here: the next round starts with one, after the accumulators have gone to sC)
  }
  if (!s_ok && tid == 0) {   // a dependency never came: let everybody run out
    ct_raise(failflag);
    for (int t = 0; t < nT * (nT + 1) / 2; ++t) ct_raise(C.flag + t);
  }
}

// one workgroup per lower tile, blockIdx.x in column-major task order
__device__ void chol_tile_task(const CholTiles& C, int task, double* lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nT = C.nT;
  if (task >= nT * (nT + 1) / 2) {   // the back-substitution tasks sit behind the tiles, last tile row first
    chol_backsub_task(C, nT - 1 - (task - nT * (nT + 1) / 2), lds);
    return;
  }
  if (task == 0) {
    chol_chain_task(C, lds);
    return;
  }
  int j = 0, rem = task;
  while (rem >= nT - j) {
    rem -= nT - j;
    ++j;
  }
  const int i = j + rem;
  double* sA = lds;                       // operand / work tiles
  double* sB = lds + CT_TB * CT_LD;
  __shared__ int s_ok;
  __shared__ double s_r[CT_TB];
  int* failflag = C.flag + nT * (nT + 1) / 2;
  if (tid == 0) s_ok = 1;
  const bool diag = (i == j);        // (j >= 1) prepares tile j of the chain: everything but the last update
  const bool sub = (i == j + 1);     // prepares C2_(j+1,j) for the chain
  const int kend = diag ? j - 1 : j;
  if (C.tl && tid == 0) C.tl[4 * task] = (double)wall_clock64();
  ct_v4 acc[3];
  double* Tij = C.T + (size_t)ct_tile_index(i, j) * CT_TILE;
  if (wave < 3) ct_load_acc_g(acc, Tij, wave, lane);
  if (diag && tid < CT_TB) s_r[tid] = C.rhs[CT_TB * j + tid];
  __syncthreads();
  for (int k = 0; k < kend; ++k) {
    if (tid == 0) {
      bool ok = ct_wait_col(C, C.flag + ct_tile_index(i, k), k);
      if (ok && !diag) ok = ct_wait(C.flag + ct_tile_index(j, k));
      if (!ok) s_ok = 0;
    }
    __syncthreads();
    if (!s_ok) break;
    ct_load_tile(C.T + (size_t)ct_tile_index(i, k) * CT_TILE, sA, tid);
    if (!diag) ct_load_tile(C.T + (size_t)ct_tile_index(j, k) * CT_TILE, sB, tid);
    __syncthreads();
    if (wave < 3) ct_gemm_nt(acc, sA, diag ? sA : sB, wave, lane, -1.0);
    if (diag && tid >= 192 && tid < 192 + CT_TB) {  // forward substitution rides along: r_j -= L_jk y_k
      const int r = tid - 192;
      const double* yk = C.y + CT_TB * k;
      double v = 0;
      for (int m = 0; m < CT_TB; ++m) v += sA[r * CT_LD + m] * ct_gld(yk + m);
      s_r[r] -= v;
    }
    __syncthreads();
  }
  if (!s_ok) {
    if (tid == 0) {
      ct_raise(failflag);
      if (diag) ct_raise(C.pflag + j);          // let the dependants run out
      else if (sub) ct_raise(C.pflag + nT + j);
      else ct_raise(C.flag + ct_tile_index(i, j));
    }
    return;
  }
  if (diag || sub) {
    // hand the partial to the chain workgroup (in place)
    if (wave < 3) ct_store_acc_g(acc, Tij, wave, lane);
    if (diag && tid < CT_TB) ct_gst(C.rhs + CT_TB * j + tid, s_r[tid]);
    ct_release();
    __syncthreads();
    if (tid == 0) ct_raise(C.pflag + (diag ? j : nT + j));
    return;
  }
  // L_ij = C Linv_j^T
  if (wave < 3) ct_store_acc(acc, sA, CT_LD, wave, lane);
  if (tid == 0 && !ct_wait_col(C, C.flag + ct_tile_index(j, j), j)) s_ok = 0;
  if (C.tl && tid == 0) C.tl[4 * task + 1] = (double)wall_clock64();
  __syncthreads();
  ct_load_tile(C.Linv + (size_t)j * CT_TILE, sB, tid);
  __syncthreads();
  if (wave < 3) {
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] = ct_v4{0.0, 0.0, 0.0, 0.0};
    ct_gemm_nt(acc, sA, sB, wave, lane, 1.0);
    ct_store_acc_g(acc, Tij, wave, lane);
  }
  if (tid == 0 && !s_ok) ct_raise(failflag);
  ct_release();
  __syncthreads();
  if (tid == 0) ct_raise(C.flag + ct_tile_index(i, j));
  if (C.tl && tid == 0) C.tl[4 * task + 3] = (double)wall_clock64();
}

constexpr int CT_SMEM_DOUBLES = 4 * CT_TB * CT_LD;   // (the chain workgroup keeps four tiles)

// stand-alone solve of one dense SPD system (tests / diagnostics): grid.x = number of lower tiles + nT
__global__ __launch_bounds__(CT_THREADS) void chol_tile_kernel(CholTiles C) {
  extern __shared__ __attribute__((aligned(16))) double ct_smem[];
  chol_tile_task(C, blockIdx.x, ct_smem);
}
}  // namespace ba
