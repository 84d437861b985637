// TV primitives on gfx950: the Chambolle dual-projection iteration (K1), the
// final f = g - lambda div p (K2) and the periodic isotropic TV norm (K3).
//
// Arithmetic follows utils/chambolle_prox_TV_stop.m:120-166 of the reference
// operation for operation (fp contraction is disabled in this file so a
// multiply-add sequence rounds exactly like MATLAB's separate operations).
//
// Data layout: column-major, i (row) is the contiguous axis.  One workgroup of
// 256 threads (4 wave64) owns a tile of TI=128 rows x TJ=16 columns; lane l of
// a wave owns the row pair (2l, 2l+1) so global accesses are 16-byte (double2)
// per lane and fully coalesced.  u = div p - g/lambda is computed once per
// pixel (plus a one-row / one-column halo) into LDS, then the forward
// differences of u are read back from LDS.
#include <mutex>

#include <algorithm>
#include "sbtv_internal.h"

#pragma clang fp contract(off)

namespace sbtv {

constexpr int TI = 128;          // tile rows   (64 lanes x 2 rows)
constexpr int TJ = 16;           // tile columns (4 waves x 4 columns)
constexpr int CJ = TJ / 4;       // columns per wave
constexpr int ULD = TI + 2;      // LDS leading dimension of the u tile (even -> 16 B aligned rows)
constexpr int TVB = 256;         // threads per block

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// deterministic block sum for 256 threads; result valid in thread 0
__device__ __forceinline__ double block_sum_256(double v, double *red /*[4]*/) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = (red[0] + red[1]) + (red[2] + red[3]);
    return r;
}

// x-divergence term at row i (chambolle_prox_TV_stop.m:156-157):
//   i == 0: p(0) ; 0 < i < M-1: p(i) - p(i-1) ; i == M-1: -p(M-1)
__device__ __forceinline__ double div1(double p, double pm, int i, int n) {
    return (i == 0) ? p : ((i == n - 1) ? -p : (p - pm));
}

template <bool VEC>
__device__ __forceinline__ double2 ld2(const double *__restrict__ base, size_t off, bool ok0, bool ok1) {
    double2 r = make_double2(0.0, 0.0);
    if (VEC) {
        if (ok1) {
            r = *reinterpret_cast<const double2 *>(base + off);
        } else if (ok0) {
            r.x = base[off];
        }
    } else {
        if (ok0) r.x = base[off];
        if (ok1) r.y = base[off + 1];
    }
    return r;
}

template <bool VEC>
__device__ __forceinline__ void st2(double *__restrict__ base, size_t off, double2 v, bool ok0, bool ok1) {
    if (VEC) {
        if (ok1) {
            *reinterpret_cast<double2 *>(base + off) = v;
        } else if (ok0) {
            base[off] = v.x;
        }
    } else {
        if (ok0) base[off] = v.x;
        if (ok1) base[off + 1] = v.y;
    }
}

// One Chambolle iteration (chambolle_prox_TV_stop.m:121-130) on every image of
// the batch whose control block is not `done`.
template <bool VEC>
__global__ __launch_bounds__(TVB) void chambolle_iter_kernel(const double *__restrict__ g, double *__restrict__ pbuf,
                                                              const ProxCtrl *__restrict__ ctrl,
                                                              double *__restrict__ partials, int M, int N,
                                                              int batch, int tiles_i) {
    const int b = blockIdx.z;
    const ProxCtrl c = ctrl[b];
    if (c.done) return;
    const size_t P = (size_t)M * N;
    const size_t plane = P * batch;
    const double *__restrict__ pxi = pbuf + (size_t)((c.cur & 1) * 2 + 0) * plane + (size_t)b * P;
    const double *__restrict__ pyi = pbuf + (size_t)((c.cur & 1) * 2 + 1) * plane + (size_t)b * P;
    double *__restrict__ pxo = pbuf + (size_t)(((c.cur & 1) ^ 1) * 2 + 0) * plane + (size_t)b * P;
    double *__restrict__ pyo = pbuf + (size_t)(((c.cur & 1) ^ 1) * 2 + 1) * plane + (size_t)b * P;
    const double *__restrict__ gg = g + (size_t)b * P;
    const double lambda = c.lambda, tau = c.tau;

    __shared__ __attribute__((aligned(16))) double u_lds[(TJ + 1) * ULD];
    __shared__ double red[4];

    const int ti = blockIdx.x, tj = blockIdx.y;
    const int i0 = ti * TI, j0 = tj * TJ;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int il = 2 * lane;            // local row of the pair
    const int i = i0 + il;
    const bool ok0 = i < M, ok1 = i + 1 < M;

    double2 px[CJ], py[CJ];

    // ---- phase 1: u = div p - g/lambda on the tile, its halo column and halo row
    // halo column jl = TJ is handled by wave 0 as a fifth column
    const int ncol = CJ + ((w == 0) ? 1 : 0);
    double2 pym = make_double2(0.0, 0.0);       // py(i, j-1)
    for (int c4 = 0; c4 < ncol; ++c4) {
        const int jl = (c4 < CJ) ? (w * CJ + c4) : TJ;
        const int j = j0 + jl;
        if (j >= N) break;                       // wave-uniform
        const size_t off = (size_t)j * M + i;
        const double2 pxv = ld2<VEC>(pxi, off, ok0, ok1);
        const double2 pyv = ld2<VEC>(pyi, off, ok0, ok1);
        const double2 gv = ld2<VEC>(gg, off, ok0, ok1);
        if (c4 == 0 || c4 == CJ) {
            pym = (j > 0) ? ld2<VEC>(pyi, off - M, ok0, ok1) : make_double2(0.0, 0.0);
        }
        // px(i-1, j): odd row of the previous lane, or a scalar load at the tile edge
        double pxm = __shfl_up(pxv.y, 1, 64);
        if (lane == 0) pxm = (i > 0 && ok0) ? pxi[off - 1] : 0.0;
        const double dx0 = div1(pxv.x, pxm, i, M);
        const double dx1 = div1(pxv.y, pxv.x, i + 1, M);
        const double dy0 = div1(pyv.x, pym.x, j, N);
        const double dy1 = div1(pyv.y, pym.y, j, N);
        double2 u;
        u.x = (dy0 + dx0) - gv.x / lambda;       // divp = v + u (:159), u = divp - g/lambda (:124)
        u.y = (dy1 + dx1) - gv.y / lambda;
        *reinterpret_cast<double2 *>(&u_lds[jl * ULD + il]) = u;
        if (c4 < CJ) {
            px[c4] = pxv;
            py[c4] = pyv;
        }
        pym = pyv;
    }
    // halo row il = TI (global row i0+TI) for the TJ columns: wave 1, lanes 0..TJ-1
    if (w == 1 && lane < TJ) {
        const int ih = i0 + TI;
        const int j = j0 + lane;
        if (ih < M && j < N) {
            const size_t off = (size_t)j * M + ih;
            const double pxv = pxi[off], pxm = pxi[off - 1];
            const double pyv = pyi[off];
            const double pymv = (j > 0) ? pyi[off - M] : 0.0;
            const double dx = div1(pxv, pxm, ih, M);
            const double dy = div1(pyv, pymv, j, N);
            u_lds[lane * ULD + TI] = (dy + dx) - gg[off] / lambda;
        }
    }
    __syncthreads();

    // ---- phase 2: gradient of u, error term with the OLD p, dual update
    double esum = 0.0;
#pragma unroll
    for (int c4 = 0; c4 < CJ; ++c4) {
        const int jl = w * CJ + c4;
        const int j = j0 + jl;
        if (j < N) {
            const double2 uc = *reinterpret_cast<const double2 *>(&u_lds[jl * ULD + il]);
            const double ud = u_lds[jl * ULD + il + 2];                 // u(i+2, j)
            const double2 ur = *reinterpret_cast<const double2 *>(&u_lds[(jl + 1) * ULD + il]);
            const bool jin = (j < N - 1);
            // GradientIm (:161-166): last row of dux and last column of duy are zero
            const double upx0 = (i < M - 1) ? (uc.y - uc.x) : 0.0;
            const double upx1 = (i + 1 < M - 1) ? (ud - uc.y) : 0.0;
            const double upy0 = jin ? (ur.x - uc.x) : 0.0;
            const double upy1 = jin ? (ur.y - uc.y) : 0.0;
            const double t0 = sqrt(upx0 * upx0 + upy0 * upy0);          // :127
            const double t1 = sqrt(upx1 * upx1 + upy1 * upy1);
            if (ok0) {
                const double a = -upx0 + t0 * px[c4].x, bb = -upy0 + t0 * py[c4].x;   // :128
                esum += a * a + bb * bb;
            }
            if (ok1) {
                const double a = -upx1 + t1 * px[c4].y, bb = -upy1 + t1 * py[c4].y;
                esum += a * a + bb * bb;
            }
            double2 npx, npy;
            npx.x = (px[c4].x + tau * upx0) / (1.0 + tau * t0);         // :129
            npx.y = (px[c4].y + tau * upx1) / (1.0 + tau * t1);
            npy.x = (py[c4].x + tau * upy0) / (1.0 + tau * t0);         // :130
            npy.y = (py[c4].y + tau * upy1) / (1.0 + tau * t1);
            const size_t off = (size_t)j * M + i;
            st2<VEC>(pxo, off, npx, ok0, ok1);
            st2<VEC>(pyo, off, npy, ok0, ok1);
        }
    }
    const double bs = block_sum_256(esum, red);
    if (threadIdx.x == 0) partials[(size_t)b * (gridDim.x * gridDim.y) + (size_t)tj * tiles_i + ti] = bs;
}

// ---------------------------------------------------------------------------
// Temporally fused Chambolle kernel: up to FH iterations per launch.
//
// A workgroup (FNW wave64, default 8) owns a REGION of FRI=128 rows x FRJ = FNW*FCJ columns (default 8 x 4 = 32)
// that is the CORE tile (116 x 21) plus a halo of FH=6 rows on either side and FHL=6 / FHJ=5 columns; one
// iteration has a dependency radius of one pixel, so after s <= 5 iterations
// the core is still exact.  The whole state of the region lives in REGISTERS:
// lane l of wave w owns the row pair (2l, 2l+1) of the FCJ columns
// [FCJ*w, FCJ*w + FCJ) : px, py = 16 doubles per lane; g/lambda is parked in LDS.  Row neighbours
// come from the adjacent lanes through DPP wave shifts (no LDS, no memory);
// column neighbours are in the same lane's registers except at the FNW-1
// wave seams, which exchange one column of u / py per iteration through LDS.
// HBM traffic per launch is one read of (g,px,py) over the region and one
// write of (px,py) over the core, for up to 5 iterations of work.
// (Other geometries: SBTV_FUSED_VARIANT; measured in profiles/r02_chambolle_variants.md.)
//
// The stop rule of chambolle_prox_TV_stop.m:131 is evaluated after the launch
// from per-iteration error partials; if it fired in the middle of a launch the
// control kernel records `redo` and a (normally empty) re-run launch repeats
// exactly that many steps from the untouched input buffer, so results are
// identical to one-iteration-at-a-time execution.
// ---------------------------------------------------------------------------
constexpr int FH = 6;                    // halo = max fused iterations of the TILE kernels
// (FSMAX, the step capacity of the error partials, is defined in sbtv_internal.h: the SALSA collector reads them too)
constexpr int FHJ = 5;                   // right column halo = max steps per launch (rows need an even halo: FH)
constexpr int FHL = FHJ + 1;             // left column halo: one more, because f = g - lambda div p written by the
                                         // last launch needs py(i, j-1) of the FINAL iterate left of the core
constexpr int FRI = 128;                 // region rows
constexpr int FCI = FRI - 2 * FH;        // core rows   (116)
// one-row-per-lane tiles (tv_fused1.inc)
constexpr int F1RI = 64;                 // region rows
constexpr int F1HT = FHJ + 1;            // halo above the core (the f epilogue needs px(i-1) of the final iterate)
constexpr int F1HB = FHJ;                // halo below
constexpr int F1CI = F1RI - F1HT - F1HB; // core rows (53)
// mixed launch of the 128-row kernel: its last workgroups work on 64-row tiles (see prox_plan)
struct FusedMix {
    int nfull, nhi, row0;                // 128-row tiles (ids below nfull), tile rows of the 64-row part, its first image row
    int stagger;                         // half-microseconds the second workgroup of every CU waits at the start of a launch
};
// Stagger (round 4, profiles/r04_chambolle_stagger.md): the 512 workgroups of the first round start together, so the two
// workgroups of a CU wait for their regions at the same time and then share the vector units at the same time.  Holding
// workgroups 256..511 (the second one of every CU) back by 3 us lets one of them load while the other iterates: -3 % time per
// Chambolle iteration in the loop, +1.5 % SALSA outer iterations/s at 2048^2, same bits.  Only for grids of at least three
// rounds (a one-round grid would just end 3 us later); SBTV_FUSED_STAGGER=n: n half-microseconds, 0 = off.
static int fused_stagger(int nblk) {
    static const int v = [] {
        const char *e = getenv("SBTV_FUSED_STAGGER");
        return e ? atoi(e) : 6;
    }();
    return nblk >= 3 * 256 ? v : 0;
}
// tuning parameters: columns per wave (cj) and waves per block (nw): region columns = nw*cj,
// core columns = nw*cj - 2 FHJ; minw = waves per SIMD requested from the register allocator
struct FusedVariant { int cj; int nw; int minw; int fast; int rpl; };   // rpl = rows per lane (2: tv_fused.inc, 1: tv_fused1.inc)
static FusedVariant g_fused = {4, 8, 4, 1, 2};
static bool g_fused_forced = false;             // SBTV_FUSED_VARIANT given: no per-plan choice

__device__ __forceinline__ double dpp_from_prev_lane(double v) {   // lane l gets lane l-1 (lane 0: 0)
    int lo = __double2loint(v), hi = __double2hiint(v);
    // bound_ctrl: the lane without a source reads 0 by itself - with `false` the compiler has to write the 0 into the
    // destination first (two v_mov per 64-bit move: 16 of the 292 vector instructions of a fused iteration)
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);   // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_from_next_lane(double v) {   // lane l gets lane l+1 (lane 63: 0)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);   // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// per-workgroup error partial: write-through store (`global_store ... sc1`) so that the in-kernel
// control path can hand it to another workgroup without an L2 write-back fence
__device__ __forceinline__ void fused_store_partial(double *p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- stop rule shared by the control kernel and the in-kernel ("last workgroup") control path ----
// tots[s] = sum of the per-workgroup error partials of fused step s, in a fixed order: wave w takes
// steps w, w+nwaves, ... (lane-strided accumulation, then the xor tree)
__device__ __forceinline__ void fused_reduce_steps(const double *__restrict__ partials_b, int nblk, int nsteps,
                                                   double *__restrict__ tots /* LDS [FH] */) {
    const int nwaves = blockDim.x >> 6, lane = threadIdx.x & 63;
    for (int s = threadIdx.x >> 6; s < nsteps; s += nwaves) {
        const double *p = partials_b + (size_t)s * nblk;
        double acc = 0.0;
        for (int base = 0; base < nblk; base += 64 * 16) {           // 16 independent loads in flight
            double v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = base + r * 64 + lane;
                // agent-scope relaxed atomic load = `global_load ... sc1`: never served from a stale L1 line
                v[r] = (q < nblk) ? __hip_atomic_load(p + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc += v[r];
        }
        acc = wave_sum(acc);
        if (lane == 0) tots[s] = acc;
    }
    __syncthreads();
}

// cont = (k < MaxIter) & (err > tol)  (chambolle_prox_TV_stop.m:131), applied to the nsteps iterations
// of one fused launch; one thread calls this
__device__ __forceinline__ void fused_apply_stop_rule(ProxCtrl *c, const double *tots, int nsteps, int write_f) {
    for (int s = 0; s < nsteps; ++s) {
        const int k = c->k + s + 1;
        const double err = sqrt(tots[s]);
        const bool stop = !((k < c->maxiter) && (err > c->tol));
        if (stop) {
            c->done = 1;
            if (s == nsteps - 1) {
                c->k = k;
                c->err = err;
                c->cur ^= 1;
                if (write_f) c->f_valid = 1;   // this launch's f is the final one
            } else {
                c->redo = s + 1;               // over-ran: the output buffer is too far; re-run s+1 steps
            }
            return;
        }
    }
    c->k += nsteps;
    c->err = sqrt(tots[nsteps - 1]);
    c->cur ^= 1;
    if (write_f && c->k >= c->maxiter) c->f_valid = 1;
}

// In-kernel control: the LAST workgroup of image b to finish a fused launch reduces the partials and
// applies the stop rule, which saves a dependent control-kernel launch (~6 us) per fused launch.
// Inter-workgroup hand-off per cdna_hip_programming.md Guideline 16: every storing wave drains its
// stores, workgroup barrier, lane 0: agent-scope release -> relaxed agent atomic ticket; the workgroup
// that draws the last ticket: agent-scope acquire -> barrier -> plain loads of all partials.
__device__ __forceinline__ void fused_inline_ctrl(ProxCtrl *__restrict__ c, const double *__restrict__ partials_b,
                                                  int nblk, int nsteps, int write_f,
                                                  unsigned *__restrict__ counter_b, int redo_mode) {
    __shared__ double ic_tots[FSMAX];
    __shared__ int ic_last;
    // The partials were stored write-through (`sc1`, fused_store_partial) so no release fence (an L2
    // write-back of the megabytes of freshly written duals!) is needed: drain, barrier, ticket.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(counter_b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == gridDim.x - 1) ? 1 : 0;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        ic_last = last;
    }
    __syncthreads();
    if (!ic_last) return;
    if (redo_mode) {
        // the re-run / finish-only launch that actually had work: book its nsteps iterations (the stop rule is
        // known to end exactly there) and mark f as valid.  Only the LAST workgroup may do this: an early one
        // setting f_valid would let late workgroups of the same launch skip their tile.
        if (nsteps > 0) fused_reduce_steps(partials_b, nblk, nsteps, ic_tots);
        if (threadIdx.x == 0) {
            if (nsteps > 0) {
                c->k += nsteps;
                c->err = sqrt(ic_tots[nsteps - 1]);
                c->cur ^= 1;
                c->redo = 0;
            }
            if (write_f) c->f_valid = 1;
            *counter_b = 0;
        }
        return;
    }
    fused_reduce_steps(partials_b, nblk, nsteps, ic_tots);
    if (threadIdx.x == 0) {
        fused_apply_stop_rule(c, ic_tots, nsteps, write_f);
        *counter_b = 0;                       // ready for the next launch (kernel boundary orders it)
    }
}

#ifdef SBTV_FUSED_TIMELINE
// Debug build only (make timeline -> lib/libsbtv_timeline.so; tools/chambolle_timeline.py): thread 0 of every workgroup
// of the two-rows-per-lane kernel leaves the constant-rate clock (100 MHz) at entry, after its region has arrived, after
// its last iteration and after its stores were issued, plus where it ran; the LAST launch's records are read back with
// sbtv_debug_timeline().
__device__ unsigned long long g_fused_tl[8 * 8192];
__device__ __forceinline__ void fused_tl(int slot) {
    if (threadIdx.x != 0 || blockIdx.x >= 8192) return;
    unsigned long long *r = g_fused_tl + (size_t)blockIdx.x * 8;
    r[slot] = wall_clock64();
    if (slot == 0) {
        r[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_ID: wave / simd / cu / sh / se
        r[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);       // XCC_ID
    }
}
#define SBTV_TL(slot) fused_tl(slot)
// per-WAVE stamps of the shader-clock counter around the two barriers of every fused iteration, for every 16th workgroup:
// slot 0 = entry of the step loop, then per step: before / after the first barrier, before / after the second one
constexpr int TLW_SLOTS = 24, TLW_SAMPLES = 128, TLW_EVERY = 16;
__device__ unsigned long long g_fused_tlw[TLW_SAMPLES * 16 * TLW_SLOTS];
__device__ __forceinline__ void fused_tlw(int slot) {
    if ((threadIdx.x & 63) != 0 || (blockIdx.x % TLW_EVERY) != 0 || blockIdx.x / TLW_EVERY >= TLW_SAMPLES || slot >= TLW_SLOTS) return;
    g_fused_tlw[((size_t)(blockIdx.x / TLW_EVERY) * 16 + (threadIdx.x >> 6)) * TLW_SLOTS + slot] = __builtin_readcyclecounter();
}
#define SBTV_TLW(slot) fused_tlw(slot)
#else
#define SBTV_TL(slot)
#define SBTV_TLW(slot)
#endif
#include "tv_fused.inc"
#include "tv_fused1.inc"
#ifdef SBTV_LAB
#include "tv_pipe.inc"      // streaming pipeline kernel: measured and lost (profiles/r02_chambolle_variants.md); lab build only
#else
constexpr int PK = 10;      // (the pipeline kernel's iterations per launch; the plan code below keeps its arithmetic)
#endif

// Stop rule after a fused launch of `steps_arg` iterations (see the kernel header).
__global__ __launch_bounds__(64 * FSMAX) void chambolle_fused_ctrl_kernel(ProxCtrl *__restrict__ ctrl,
                                                                       const double *__restrict__ partials, int nblk,
                                                                       int steps_arg, int redo_mode, int write_f) {
    const int b = blockIdx.x;
    ProxCtrl *c = &ctrl[b];
    __shared__ double tots[FSMAX];
    __shared__ double part[FSMAX];
    // This kernel is a chain of dependent memory round trips (control block, partials, control block), each ~1 us
    // from another XCD's write-through data.  The partial sums of all `steps_arg` possible steps are therefore requested
    // at once and BEFORE the control block is looked at: FSMAX waves, wave w takes step w % steps_arg and the
    // (w / steps_arg)-th share of its nblk partials, every lane's loads in one batch.
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int smax = redo_mode ? FSMAX : max(1, min(steps_arg, FSMAX));
    const int shares = FSMAX / smax;                   // waves per step (>= 1)
    const int st = w % smax, sh = w / smax;
    double acc = 0.0;
    if (sh < shares) {
        const double *p = partials + ((size_t)b * FSTRIDE + st) * nblk;
        const int per = (nblk + shares - 1) / shares, q0 = sh * per, q1 = min(nblk, q0 + per);
        constexpr int NB = 32;                          // up to 32 loads per lane in flight (nblk <= 2048 x shares)
        for (int base = q0; base < q1; base += 64 * NB) {
            double v[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                const int q = base + r * 64 + lane;
                // agent-scope relaxed atomic load = `global_load ... sc1`: never served from a stale L1 line
                v[r] = (q < q1) ? __hip_atomic_load(p + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            }
#pragma unroll
            for (int r = 0; r < NB; ++r) acc += v[r];
        }
        acc = wave_sum(acc);
    }
    int nsteps;
    if (redo_mode) {
        if (c->redo <= 0) {
            if (threadIdx.x == 0 && write_f) c->f_valid = 1;   // the redo launch ran as a finish-only pass (or f was valid)
            return;
        }
        nsteps = c->redo;
    } else {
        if (c->done) return;
        nsteps = min(steps_arg, c->maxiter - c->k);
        if (nsteps <= 0) return;
    }
    if (lane == 0) part[w] = acc;
    __syncthreads();
    if ((int)threadIdx.x < nsteps) {
        double t = 0.0;
        for (int h = 0; h < shares; ++h) t += part[h * smax + threadIdx.x];     // fixed order
        tots[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (redo_mode) {
            c->k += nsteps;
            c->err = sqrt(tots[nsteps - 1]);
            c->cur ^= 1;
            c->redo = 0;
            if (write_f) c->f_valid = 1;
            return;
        }
        fused_apply_stop_rule(c, tots, nsteps, write_f);
    }
}

// Stop rule of a multi-buffer optimistic prox (prox_iterate, mode 2): the launches ran all `total` iterations of a COLD
// prox, launch l of `base` (+1 for the first `extra`) steps from dual buffer l into buffer l + 1.  One workgroup per
// image totals the error partials of every step (fixed order) and applies cont = (k < MaxIter) & (err > tol)
// (chambolle_prox_TV_stop.m:131).  Stopped at the last step: the f the last launch wrote stands (f_valid).  Stopped at
// an earlier k: `cur` = the buffer of the launch boundary before k, `redo` = the steps from there to k; the redo
// launch that follows re-runs them and rewrites f.
__global__ __launch_bounds__(64 * FSMAX) void chambolle_mb_ctrl_kernel(ProxCtrl *__restrict__ ctrl,
                                                                    const double *__restrict__ partials, int nblk, int total,
                                                                    int base, int extra) {
    const int b = blockIdx.x;
    __shared__ double tots[FSTRIDE];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (nblk <= 64 * 4) {
        // few tiles (small images, where this kernel's latency counts): the loads of ALL steps of this wave are issued
        // before the first sum, one round trip to the L2 instead of one per step; same summation order as below
        constexpr int SW = (FSTRIDE + FSMAX - 1) / FSMAX, NB = 4;
        double v[SW][NB];
#pragma unroll
        for (int i = 0; i < SW; ++i) {
            const int st = w + i * FSMAX;
            const double *p = partials + ((size_t)b * FSTRIDE + (st < total ? st : 0)) * nblk;
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                const int q = r * 64 + lane;
                v[i][r] = (st < total && q < nblk) ? __hip_atomic_load(p + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            }
        }
#pragma unroll
        for (int i = 0; i < SW; ++i) {
            const int st = w + i * FSMAX;
            double acc = 0.0;
#pragma unroll
            for (int r = 0; r < NB; ++r) acc += v[i][r];
            acc = wave_sum(acc);
            if (lane == 0 && st < total) tots[st] = acc;
        }
    } else
    for (int st = w; st < total; st += FSMAX) {
        const double *p = partials + ((size_t)b * FSTRIDE + st) * nblk;
        double acc = 0.0;
        constexpr int NB = 16;
        for (int q0 = 0; q0 < nblk; q0 += 64 * NB) {
            double v[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                const int q = q0 + r * 64 + lane;
                v[r] = (q < nblk) ? __hip_atomic_load(p + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            }
#pragma unroll
            for (int r = 0; r < NB; ++r) acc += v[r];
        }
        acc = wave_sum(acc);
        if (lane == 0) tots[st] = acc;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    ProxCtrl c = ctrl[b];
    if (c.done) return;                         // parked image
    mb_apply_rule(c, tots, total, base, extra);
    ctrl[b] = c;
}

// Sum the per-block partials in a fixed order, then apply the stop rule
// cont = (k < MaxIter) & (err > tol)   (chambolle_prox_TV_stop.m:131)
__global__ __launch_bounds__(256) void chambolle_ctrl_kernel(ProxCtrl *__restrict__ ctrl,
                                                              const double *__restrict__ partials, int nblk) {
    const int b = blockIdx.x;
    ProxCtrl *c = &ctrl[b];
    if (c->done) return;
    __shared__ double red[4];
    const double *p = partials + (size_t)b * nblk;
    double s = 0.0;
    for (int q = threadIdx.x; q < nblk; q += 256) s += p[q];
    const double tot = block_sum_256(s, red);
    if (threadIdx.x == 0) {
        const int k = c->k + 1;
        const double err = sqrt(tot);
        c->k = k;
        c->err = err;
        c->cur ^= 1;
        c->done = !((k < c->maxiter) && (err > c->tol));
    }
}

__global__ void prox_reset_kernel(ProxCtrl *__restrict__ ctrl, const double *__restrict__ lambda,
                                  double lambda_scale, int maxiter, double tol, double tau, int keep_cur,
                                  int batch, const int *__restrict__ frozen) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    ProxCtrl c = ctrl[b];
    c.k = 0;
    c.done = (frozen && frozen[b]) ? 1 : 0;
    c.cur = keep_cur ? (c.cur & 1) : 0;      // never anything but 0 or 1, whatever the buffer held
    c.maxiter = maxiter;
    c.redo = 0;
    c.f_valid = 0;
    c.err = 0.0;
    c.lambda = lambda[b] * lambda_scale;
    c.tol = tol;
    c.tau = tau;
    ctrl[b] = c;
}

// f = g - lambda * DivergenceIm(px, py)   (chambolle_prox_TV_stop.m:149)
template <bool VEC>
__global__ __launch_bounds__(TVB) void chambolle_finish_kernel(const double *__restrict__ g,
                                                                const double *__restrict__ pbuf,
                                                                const ProxCtrl *__restrict__ ctrl,
                                                                double *__restrict__ f, int M, int N, int batch) {
    const int b = blockIdx.z;
    const ProxCtrl c = ctrl[b];
    const size_t P = (size_t)M * N;
    const size_t plane = P * batch;
    const double *__restrict__ px = pbuf + (size_t)((c.cur & 1) * 2 + 0) * plane + (size_t)b * P;
    const double *__restrict__ py = pbuf + (size_t)((c.cur & 1) * 2 + 1) * plane + (size_t)b * P;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * TI + 2 * lane;
    const bool ok0 = i < M, ok1 = i + 1 < M;
    const double lambda = c.lambda;
#pragma unroll
    for (int c4 = 0; c4 < CJ; ++c4) {
        const int j = blockIdx.y * TJ + w * CJ + c4;
        if (j >= N) break;
        const size_t off = (size_t)j * M + i;
        const double2 pxv = ld2<VEC>(px, off, ok0, ok1);
        const double2 pyv = ld2<VEC>(py, off, ok0, ok1);
        const double2 gv = ld2<VEC>(g + (size_t)b * P, off, ok0, ok1);
        const double2 pym = (j > 0) ? ld2<VEC>(py, off - M, ok0, ok1) : make_double2(0.0, 0.0);
        double pxm = __shfl_up(pxv.y, 1, 64);
        if (lane == 0) pxm = (i > 0 && ok0) ? px[off - 1] : 0.0;
        double2 r;
        r.x = gv.x - lambda * (div1(pyv.x, pym.x, j, N) + div1(pxv.x, pxm, i, M));
        r.y = gv.y - lambda * (div1(pyv.y, pym.y, j, N) + div1(pxv.y, pxv.x, i + 1, M));
        st2<VEC>(f + (size_t)b * P, off, r, ok0, ok1);
    }
}

// Periodic isotropic TV (utils/TVnorm.m:2, SALSA/diffh.m, diffv.m):
//   sum sqrt( (x(i,j)-x(i,j-1))^2 + (x(i,j)-x(i-1,j))^2 ), indices modulo the size
template <bool VEC>
__global__ __launch_bounds__(TVB) void tvnorm_kernel(const double *__restrict__ x, double *__restrict__ partials,
                                                      int M, int N, int tiles_i) {
    const int b = blockIdx.z;
    const size_t P = (size_t)M * N;
    const double *__restrict__ xb = x + (size_t)b * P;
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * TI + 2 * lane;
    const bool ok0 = i < M, ok1 = i + 1 < M;
    double s = 0.0;
#pragma unroll
    for (int c4 = 0; c4 < CJ; ++c4) {
        const int j = blockIdx.y * TJ + w * CJ + c4;
        if (j >= N) break;
        const size_t off = (size_t)j * M + i;
        const int jm = (j > 0) ? j - 1 : N - 1;
        const double2 xv = ld2<VEC>(xb, off, ok0, ok1);
        const double2 xl = ld2<VEC>(xb, (size_t)jm * M + i, ok0, ok1);
        double xu = __shfl_up(xv.y, 1, 64);
        if (lane == 0 && ok0) xu = xb[(size_t)j * M + ((i > 0) ? i - 1 : M - 1)];
        if (ok0) {
            const double dh = xv.x - xl.x, dv = xv.x - xu;
            s += sqrt(dh * dh + dv * dv);
        }
        if (ok1) {
            const double dh = xv.y - xl.y, dv = xv.y - xv.x;
            s += sqrt(dh * dh + dv * dv);
        }
    }
    const double bs = block_sum_256(s, red);
    if (threadIdx.x == 0) partials[(size_t)b * (gridDim.x * gridDim.y) + (size_t)blockIdx.y * tiles_i + blockIdx.x] = bs;
}

// out[v] = sum_q partials[v*n + q], fixed order
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double *__restrict__ partials, int n,
                                                               double *__restrict__ out) {
    __shared__ double red[4];
    const double *p = partials + (size_t)blockIdx.x * n;
    double s = 0.0;
    for (int q = threadIdx.x; q < n; q += 256) s += p[q];
    const double tot = block_sum_256(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = tot;
}

// the same for up to four sets of vectors in ONE launch (a solver loop ends an iteration with several small reductions,
// each a launch of a few microseconds on an otherwise idle GPU); same summation order as reduce_partials_kernel
__global__ __launch_bounds__(256) void reduce_jobs_kernel(RedJobs jb) {
    __shared__ double red[4];
    int v = blockIdx.x, q = 0;
    while (q < 3 && v >= jb.nvec[q]) v -= jb.nvec[q++];
    const int n = jb.n[q];
    const double *p = jb.src[q] + (size_t)v * n;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += p[i];
    const double tot = block_sum_256(s, red);
    if (threadIdx.x == 0) jb.dst[q][v] = tot;
}

__global__ void copy_duals_kernel(double *__restrict__ dst0, double *__restrict__ dst1,
                                  const double *__restrict__ pbuf, const ProxCtrl *__restrict__ ctrl, size_t P,
                                  int batch) {
    const int b = blockIdx.y;
    const ProxCtrl c = ctrl[b];
    const size_t plane = P * batch;
    const double *px = pbuf + (size_t)((c.cur & 1) * 2 + 0) * plane + (size_t)b * P;
    const double *py = pbuf + (size_t)((c.cur & 1) * 2 + 1) * plane + (size_t)b * P;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < P; q += (size_t)gridDim.x * blockDim.x) {
        dst0[(size_t)b * P + q] = px[q];
        dst1[(size_t)b * P + q] = py[q];
    }
}

// --------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------

static inline bool vec_ok(const void *p, int M) { return (M % 2 == 0) && ((reinterpret_cast<uintptr_t>(p) & 15) == 0); }

int prox_plan(sbtv_ctx *ctx, int M, int N, int batch, ProxPlan *pl, const char *tag) {
    pl->M = M;
    pl->N = N;
    pl->batch = batch;
    pl->tiles_i = (M + TI - 1) / TI;
    pl->tiles_j = (N + TJ - 1) / TJ;
    pl->nblk = pl->tiles_i * pl->tiles_j;
    {
        // environment hooks are parsed once; contexts may be used from several host threads at the same time
        static std::once_flag env_once;
        std::call_once(env_once, [] {
            if (const char *e = getenv("SBTV_FUSED_VARIANT")) {   // tuning hook: "cj,nw,minw[,rows_per_lane]"
                int cj = 0, nw = 0, mw = 0, rpl = 2;
#ifdef SBTV_LAB
                static const int known[][4] = {{8, 4, 2, 2}, {8, 8, 2, 2}, {8, 8, 1, 2}, {8, 4, 3, 2}, {6, 8, 3, 2}, {12, 4, 2, 2}, {16, 4, 1, 2},
                                               {4, 8, 2, 2}, {4, 16, 2, 2}, {8, 6, 2, 2}, {4, 8, 3, 2}, {4, 8, 4, 2},
                                               {6, 8, 4, 2}, {6, 8, 2, 2}, {5, 8, 4, 2},
                                               {4, 8, 6, 1}, {4, 8, 5, 1}, {4, 8, 4, 1}, {8, 4, 6, 1}, {8, 8, 4, 1},
                                               {8, 4, 4, 1}, {6, 8, 6, 1}, {6, 8, 4, 1}};
#else
                // the default build carries the two geometries the plans choose themselves (128-row tiles, and 64-row
                // tiles for small grids); the others lost on MI355X and live in the lab build (make lab)
                static const int known[][4] = {{4, 8, 4, 2}, {4, 8, 4, 1}};
#endif
                bool ok = false;
                if (sscanf(e, "%d,%d,%d,%d", &cj, &nw, &mw, &rpl) >= 3)
                    for (auto &k4 : known) ok = ok || (k4[0] == cj && k4[1] == nw && k4[2] == mw && k4[3] == rpl);
                if (ok) {
                    g_fused_forced = true;
                    g_fused.cj = cj;
                    g_fused.nw = nw;
                    g_fused.minw = mw;
                    g_fused.rpl = rpl;
                }
            }
            // SBTV_EXACT=1: IEEE div/sqrt, no FMA contraction (validation build of the arithmetic)
            if (getenv("SBTV_EXACT") != nullptr) g_fused.fast = 0;
        });
    }
    // tile geometry of this plan: the process-wide variant, except that a grid which would not even give every CU
    // one 128-row tile (e.g. 512^2: 125 tiles) takes the 64-row one-row-per-lane tiles instead (twice the
    // workgroups; measured -7 % per SAPG iteration at 512^2), unless a variant was forced
    auto geometry = [&](int cj, int nw, int rpl) {
        const int core_rows = (rpl == 1) ? F1CI : FCI, core_cols = cj * nw - FHL - FHJ;
        pl->ftiles_i = (M + core_rows - 1) / core_rows;
        pl->ftiles_j = (N + core_cols - 1) / core_cols;
        pl->fnblk = pl->ftiles_i * pl->ftiles_j;
        pl->cj = cj;
        pl->nw = nw;
        pl->rpl = rpl;
    };
    geometry(g_fused.cj, g_fused.nw, g_fused.rpl);
    pl->minw = g_fused.minw;
    if (!g_fused_forced && (size_t)pl->fnblk * batch < 256) {
        geometry(4, 8, 1);
        pl->minw = 4;
    }
    // Mixed tiling of ONE large image on the shipped geometry (lab build, SBTV_TAIL_HALF=1).  A launch of T 128-row tiles runs
    // as T / S rounds of S = 2 workgroups x CUs, and in its last ~11 us (one workgroup life) the slots run empty one by one.
    // The idea: as many 128-row tile rows as fill whole rounds and the rest of the image in 64-row tiles (half the work per
    // workgroup, dispatched last through the order table), so that the slots run empty over half that time.  Same pixels,
    // same arithmetic per pixel; only the grouping of the error partials changes.  Measured: 63-65 us per launch against 60
    // with 196, 392 or 588 half tiles (profiles/r03_chambolle_tail.md): the one-row-per-lane body costs more per pixel
    // than the shorter tail gives back.  Kept in the lab build as the measured alternative.
    pl->mix_nfull = pl->mix_nfi = pl->mix_nhi = pl->mix_row0 = 0;
    pl->esub_off = 0;
#ifdef SBTV_LAB
    {
        static const bool env_on = [] {
            const char *e = getenv("SBTV_TAIL_HALF");
            return e && e[0] == '1';
        }();
        static const bool order_off = [] {
            const char *e = getenv("SBTV_TILE_ORDER");
            return e && e[0] == '0';
        }();
        const int slots = 2 * ctx->cu_count;
        if (env_on && !order_off && !g_fused_forced && batch == 1 && pl->rpl == 2 && pl->cj == 4 && pl->nw == 8 && pl->minw == 4 &&
            (M % 2 == 0) && slots > 0 && pl->fnblk > 2 * 256) {
            const int tj_n = pl->ftiles_j;
            int nfi = ((pl->fnblk / slots) * slots) / tj_n;           // tile rows that fit the complete rounds
            if (nfi > pl->ftiles_i - 1) nfi = pl->ftiles_i - 1;
            static const int env_rows = [] {                          // SBTV_TAIL_ROWS=r: the last r 128-row tile rows instead
                const char *e = getenv("SBTV_TAIL_ROWS");
                return e ? atoi(e) : 0;
            }();
            if (env_rows > 0) nfi = pl->ftiles_i - env_rows;
            const int row0 = nfi * FCI;
            if (nfi >= 1 && row0 < M) {
                const int nhi = (M - row0 + F1CI - 1) / F1CI;
                pl->mix_nfi = nfi;
                pl->mix_nhi = nhi;
                pl->mix_row0 = row0;
                pl->mix_nfull = nfi * tj_n;
                pl->fnblk = pl->mix_nfull + nhi * tj_n;
            }
        }
    }
#endif
    // Streaming pipeline kernel (tv_pipe.inc): bands of PCI core rows x column segments, one workgroup per CU.
    // OPT-IN (SBTV_PROX_PIPE=1, any even M): on MI355X it does 1.34 x instead of 1.68 x the arithmetic and a third of
    // the memory traffic, but one barrier per column step with ~100 instructions of work per wave in between leaves
    // the vector units idle half of the time: 21.1 vs 14.7 us per Chambolle iteration at 2048^2
    // (profiles/r02_chambolle_variants.md).  Kept as the measured alternative, exercised by the parity suite.
    pl->pipe = 0;
    {
        static const int env_pipe = [] {
            const char *e = getenv("SBTV_PROX_PIPE");
            return e ? atoi(e) : -1;
        }();
#ifdef SBTV_LAB
        const int nbands = (M + PCI - 1) / PCI;
#else
        const int nbands = (M + 105) / 106;
#endif
        int nseg = 256 / (nbands * batch);
        if (nseg < 1) nseg = 1;
        if (nseg > N / 80) nseg = N / 80;
        if (nseg < 1) nseg = 1;
        const int seglen = (N + nseg - 1) / nseg;
        nseg = (N + seglen - 1) / seglen;
#ifndef SBTV_LAB
        (void)env_pipe;
        if (false) {
#else
        if (env_pipe == 1 && (M % 2 == 0) && !g_fused_forced) {
#endif
            pl->pipe = 1;
            pl->nbands = nbands;
            pl->nseg = nseg;
            pl->seglen = seglen;
            pl->fnblk = nbands * nseg;
        }
    }
    const size_t P = (size_t)M * N;
    size_t npart = (size_t)batch * pl->nblk;
    if ((size_t)batch * FSTRIDE * pl->fnblk > npart) npart = (size_t)batch * FSTRIDE * pl->fnblk;
    const std::string t(tag ? tag : "prox");          // a second concurrent prox (CoRAL) needs its own state
    SBTV_TRY(ws_get_t(ctx, (t + ".ctrl").c_str(), (size_t)batch, &pl->ctrl));
    SBTV_TRY(ws_get_t(ctx, (t + ".pbuf").c_str(), 4 * P * batch, &pl->pbuf));
    pl->pairs = 2;
    pl->tag = t;
    // two sets: a solver loop whose collector rides on the next iteration's first launch reads one while that
    // launch already writes the other (salsa.hip)
    SBTV_TRY(ws_get_t(ctx, (t + ".partials").c_str(), 2 * npart, &pl->partials));
    pl->part_stride = npart;
    // arrival tickets of the in-kernel control path: zero between launches (the last workgroup resets it)
    // workgroup -> tile table of the 128-row tile kernel on grids of more than one round of workgroups: the XCD chunks
    // of the arithmetic order (workgroup id mod 8 names the XCD, each XCD walks a contiguous chunk of the tile list), but
    // inside a chunk the border tiles (boundary selects: ~1.35 x the instructions of an interior tile) come first
    pl->order = nullptr;
    {
        static const bool env_off = [] {
            const char *e = getenv("SBTV_TILE_ORDER");
            return e && e[0] == '0';
        }();
        const int nt = pl->fnblk;
        if (!env_off && !pl->pipe && pl->rpl == 2 && nt > 2 * 256) {
            int *od = nullptr;
            // one table per shape, built once per context (the name carries the shape)
            const std::string name = "prox.order." + std::to_string(M) + "x" + std::to_string(N) + "." + std::to_string(pl->cj) +
                                     "." + std::to_string(pl->nw) + (pl->mix_nfull ? ".mix" + std::to_string(pl->mix_nfi) : std::string());
            const bool built = ctx->ws.find(name) != ctx->ws.end();
            SBTV_TRY(ws_get_t(ctx, name.c_str(), (size_t)nt, &od));
            if (!built) {
                std::vector<int> h(nt);
                const int q8 = nt >> 3, r8 = nt & 7, tj_n = pl->ftiles_j;
                // 128-row tiles: ids 0 .. nf-1 on ti_n tile rows; mixed tiling: 64-row tiles nf .. nt-1 on mix_nhi tile rows
                const int nf = pl->mix_nfull ? pl->mix_nfull : nt, ti_n = pl->mix_nfull ? pl->mix_nfi : pl->ftiles_i;
                const int fq8 = nf >> 3, fr8 = nf & 7;
                int half_next = nf;
                for (int x = 0; x < 8; ++x) {
                    // this XCD's workgroup ids are o * 8 + x, o = 0 .. cnt-1: first its contiguous chunk of the 128-row tiles
                    // (border tiles before interior ones), then a contiguous run of 64-row tiles (roughly the same columns)
                    const int cnt = q8 + (x < r8 ? 1 : 0);
                    const int c0 = (x < fr8 ? x * (fq8 + 1) : fr8 * (fq8 + 1) + (x - fr8) * fq8);
                    int len = fq8 + (x < fr8 ? 1 : 0);
                    if (len > cnt) len = cnt;
                    int o = 0;
                    // experiment (SBTV_TILE_ORDER=2 / 3): inside a chunk walk the tiles ROW by row (horizontal neighbours, which
                    // share 11 of their 32 region columns, become consecutive workgroups of the XCD; by default vertical
                    // neighbours are, which share 12 of 128 rows); 3: the same without the border-first pass
                    static const int env_mode = [] {
                        const char *e = getenv("SBTV_TILE_ORDER");
                        return e ? atoi(e) : 1;
                    }();
                    std::vector<int> chunk;
                    for (int t2 = c0; t2 < c0 + len; ++t2) chunk.push_back(t2);
                    if (env_mode >= 2)
                        std::stable_sort(chunk.begin(), chunk.end(), [&](int a, int b2) { return (a % ti_n) < (b2 % ti_n); });
                    for (int pass = 0; pass < 2; ++pass)
                        for (int t2 : chunk) {
                            const int ti = t2 % ti_n, tj = t2 / ti_n;
                            const bool border = (ti == 0 || (!pl->mix_nfull && ti == ti_n - 1) || tj == 0 || tj == tj_n - 1);
                            if (env_mode == 3 ? (pass == 0) : (border == (pass == 0))) h[(size_t)(o++) * 8 + x] = t2;      // workgroup id = o * 8 + x
                        }
                    while (o < cnt) h[(size_t)(o++) * 8 + x] = half_next++;
                }
                SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
                SBTV_HIP(ctx, hipMemcpy(od, h.data(), sizeof(int) * nt, hipMemcpyHostToDevice));
            }
            pl->order = od;
        }
    }
    SBTV_TRY(ws_get_t(ctx, (t + ".counters").c_str(), (size_t)batch, &pl->counters));
    SBTV_HIP(ctx, hipMemsetAsync(pl->counters, 0, sizeof(unsigned) * batch, ctx->stream));
    return 0;
}

// more dual buffers (pairs of px, py planes) for the multi-buffer optimistic mode: launches + 2
int prox_reserve_pairs(sbtv_ctx *ctx, ProxPlan *pl, int pairs) {
    if (pairs <= pl->pairs) return 0;
    if (pairs > 15) return fail(ctx, SBTV_ERR_BADARG, "prox_reserve_pairs: at most 15 dual buffers");
    const size_t P = (size_t)pl->M * pl->N;
    SBTV_TRY(ws_get_t(ctx, (pl->tag + ".pbuf").c_str(), (size_t)2 * pairs * P * pl->batch, &pl->pbuf));
    pl->pairs = pairs;
    return 0;
}

int prox_reset(sbtv_ctx *ctx, const ProxPlan &pl, const double *lambda_dev, double lambda_scale, int maxiter,
               double tol, double tau, bool keep_cur, const int *frozen) {
    const int thr = 64;
    hipLaunchKernelGGL(prox_reset_kernel, dim3((pl.batch + thr - 1) / thr), dim3(thr), 0, ctx->stream, pl.ctrl,
                       lambda_dev, lambda_scale, maxiter, tol, tau, keep_cur ? 1 : 0, pl.batch, frozen);
    SBTV_HIP(ctx, hipGetLastError());
    return 0;
}

int prox_zero_duals(sbtv_ctx *ctx, const ProxPlan &pl) {
    const size_t P = (size_t)pl.M * pl.N;
    // only ping-pong slot 0 needs clearing (reset with keep_cur=false selects it)
    SBTV_HIP(ctx, hipMemsetAsync(pl.pbuf, 0, 2 * P * pl.batch * sizeof(double), ctx->stream));
    return 0;
}

int prox_set_duals(sbtv_ctx *ctx, const ProxPlan &pl, const double *px, const double *py) {
    const size_t P = (size_t)pl.M * pl.N * pl.batch;
    SBTV_HIP(ctx, hipMemcpyAsync(pl.pbuf, px, P * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    SBTV_HIP(ctx, hipMemcpyAsync(pl.pbuf + P, py, P * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

int prox_get_duals(sbtv_ctx *ctx, const ProxPlan &pl, double *px, double *py) {
    const size_t P = (size_t)pl.M * pl.N;
    hipLaunchKernelGGL(copy_duals_kernel, dim3(256, pl.batch), dim3(256), 0, ctx->stream, px, py, pl.pbuf, pl.ctrl,
                       P, pl.batch);
    SBTV_HIP(ctx, hipGetLastError());
    return 0;
}

int g_force_single_step = 0;   // test hook (SBTV_SINGLE_STEP=1): one-iteration kernels only

int prox_finish(sbtv_ctx *ctx, const ProxPlan &pl, const double *g, double *f);

// Can prox_iterate run optimistically (spec_cur != nullptr) for this plan and these buffers?  Needs the tile kernels
// (even M, aligned buffers, not the single-step / pipeline variants) and all steps in the partials' FSTRIDE slots.
bool prox_spec_ok(const ProxPlan &pl, const double *g, const double *f_out, int maxiter) {
    static const bool env_single = (getenv("SBTV_SINGLE_STEP") != nullptr);
    static const bool env_off = [] {
        const char *e = getenv("SBTV_PROX_SPEC");
        return e && e[0] == '0';
    }();
    return !env_off && !env_single && !g_force_single_step && !pl.pipe && maxiter <= FSTRIDE && vec_ok(g, pl.M) &&
           vec_ok(pl.pbuf, pl.M) && (!f_out || vec_ok(f_out, pl.M));
}

// number of fused launches prox_iterate makes for `maxiter` iterations (without the redo pass)
int prox_launches(const ProxPlan &pl, int maxiter) {
    const int per_launch = pl.pipe ? PK : FHJ;
    return (maxiter + per_launch - 1) / per_launch;
}

// spec: optimistic mode for solver loops (see the kernels, bit 2 of their flag word): the launches run all `maxiter`
// iterations back to back, launch l from dual buffer cur ^ (l & 1), WITHOUT stop-rule kernels and without the redo
// pass; the error partials of step s land in slot s and the control blocks are left untouched.  The caller evaluates
// the rule over the maxiter steps and re-runs exactly if it fired early; spec_parity = parity of the optimistic launches
// made since `cur` was last written (the kernels read cur ^ parity); `side`: a collector to host in the first launch.
int prox_iterate(sbtv_ctx *ctx, const ProxPlan &pl, const double *g, int maxiter, double *f_out, bool cold, int spec,
                 int spec_parity, const SideJob *side) {
    // spec: 0 exact launches, 1 optimistic ping-pong launches (warm or cold; the caller checks the rule), 2 multi-buffer
    // optimistic launches of a cold prox with the rule applied and honoured right after them (see the kernels, bit 3)
    // 3 / 4: the multi-buffer mode in two parts for a caller that applies the rule itself between them (mb_apply_rule
    // on the step sums; the SAPG loop does it in kernels it launches anyway): 3 = the launches only, 4 = the redo only
    const bool spec_cur = spec != 0;
    const bool mb = spec >= 2;
    const bool mb_main = spec != 4, mb_ctrl = spec == 2, mb_redo = spec == 2 || spec == 4;
    if (mb && (!cold || !f_out || (size_t)pl.pairs < (size_t)prox_launches(pl, maxiter) + 2))
        return fail(ctx, SBTV_ERR_BADARG, "prox_iterate: the multi-buffer mode needs a cold prox with f output and prox_reserve_pairs()");
    const dim3 grid(pl.tiles_i, pl.tiles_j, pl.batch);
    const bool v = vec_ok(g, pl.M) && vec_ok(pl.pbuf, pl.M);
    static const bool env_single = (getenv("SBTV_SINGLE_STEP") != nullptr);
    if (spec_cur && !prox_spec_ok(pl, g, f_out, maxiter))
        return fail(ctx, SBTV_ERR_BADARG, "prox_iterate: optimistic mode is not available for this plan");
    if (v && !env_single && !g_force_single_step && (!f_out || vec_ok(f_out, pl.M))) {
        // temporally fused path: ceil(K/FH) launches of (nearly) equal step counts + the redo pair
        dim3 fgrid(pl.fnblk, 1, pl.batch);         // linear tile list, remapped per XCD inside the kernel
        const SideJob no_side{};
        const int per_launch = pl.pipe ? PK : FHJ;
        const int nl = (maxiter + per_launch - 1) / per_launch;
        const int base = maxiter / nl, extra = maxiter % nl;
        // In-kernel stop rule: the last workgroup of a normal launch applies the rule itself instead of a separate
        // control kernel.  On large grids the gain is marginal (+0.6 % SALSA it/s at 2048^2), so there it needs
        // SBTV_INLINE_CTRL=1; grids below one round of workgroups (e.g. 512^2: 125 tiles) are bound by the chain
        // of dependent launches and gain ~6 % per SAPG iteration, so they take it by default (SBTV_INLINE_CTRL=0
        // switches it off).
        static const char *env_inl = getenv("SBTV_INLINE_CTRL");
        const bool env_inline = env_inl ? (env_inl[0] != '0') : ((size_t)pl.fnblk * pl.batch <= 512);
        int spec_off = 0, spec_l = 0;                           // steps / launches already made (optimistic mode)
        auto launch_fused = [&](int steps, int redo, int write_f) -> int {
            bool launched = false;
            // the first optimistic launch can host the collector of the previous outer iteration
            const SideJob &sj = (spec_cur && side && spec_l == 0) ? *side : no_side;
            fgrid.x = pl.fnblk + sj.nblocks;
            const int inl = (!redo && env_inline && !spec_cur) ? 1 : 0;
            int kflags = inl | (cold ? 2 : 0);                  // bit 0: in-kernel stop rule, bit 1: cold start
            if (spec_cur) {                                      // bit 2 + source / destination buffer + first step slot
                if (mb && redo) {
                    kflags = 4 | 8 | 2 | ((nl + 1) << 24);       // from the buffer the stop-rule kernel names into the spare one
                } else if (mb) {
                    kflags = 4 | 8 | 2 | (spec_off << 12) | (spec_l << 20) | ((spec_l + 1) << 24);
                } else {
                    kflags = 4 | (cold ? 2 : 0) | ((spec_l & 1) << 8) | ((spec_parity & 1) << 9) | (spec_off << 12);
                }
                if (!redo) {
                    ++spec_l;
                    spec_off += steps;
                }
            }
#ifdef SBTV_LAB
            if (pl.pipe) {
                launched = true;
                if (g_fused.fast)
                    hipLaunchKernelGGL(chambolle_pipe_kernel<true>, fgrid, dim3(64 * PNW), 0, ctx->stream, g, pl.pbuf,
                                       pl.ctrl, pl.partials, pl.M, pl.N, pl.batch, pl.nbands, pl.nseg, pl.seglen, pl.fnblk,
                                       steps, redo, f_out, write_f, pl.counters, kflags);
                else
                    hipLaunchKernelGGL(chambolle_pipe_kernel<false>, fgrid, dim3(64 * PNW), 0, ctx->stream, g, pl.pbuf,
                                       pl.ctrl, pl.partials, pl.M, pl.N, pl.batch, pl.nbands, pl.nseg, pl.seglen, pl.fnblk,
                                       steps, redo, f_out, write_f, pl.counters, kflags);
            }
#endif
#define SBTV_FUSED_CASE(CJ_, NW_, MW_)                                                                               \
    if (!pl.pipe && pl.rpl == 2 && pl.cj == CJ_ && pl.nw == NW_ && pl.minw == MW_) {             \
        launched = true;                                                                                             \
        if (g_fused.fast)                                                                                            \
            hipLaunchKernelGGL((chambolle_fused_kernel<CJ_, NW_, MW_, true>), fgrid, dim3(64 * NW_), 0, ctx->stream, \
                               g, pl.pbuf, pl.ctrl, pl.partials, pl.M, pl.N, pl.batch, pl.ftiles_i, pl.fnblk, steps, \
                               redo, f_out, write_f, pl.counters, kflags, sj, pl.order, FusedMix{0, 0, 0, fused_stagger(pl.fnblk)});  \
        else                                                                                                         \
            hipLaunchKernelGGL((chambolle_fused_kernel<CJ_, NW_, MW_, false>), fgrid, dim3(64 * NW_), 0,             \
                               ctx->stream, g, pl.pbuf, pl.ctrl, pl.partials, pl.M, pl.N, pl.batch, pl.ftiles_i,     \
                               pl.fnblk, steps, redo, f_out, write_f, pl.counters, kflags, sj, pl.order, FusedMix{0, 0, 0, fused_stagger(pl.fnblk)}); \
    }
            // optimistic ping-pong launches of a solver loop (spec == 1: the caller only CHECKS the rule afterwards): the error
            // sums run over a quarter of the columns (tv_fused.inc, ESUB); SBTV_ERR_SUBSET=0: all columns
            static const bool esub_wanted = [] {
                const char *e = getenv("SBTV_ERR_SUBSET");
                return !(e && e[0] == '0');
            }();
            if (esub_wanted && !pl.esub_off && spec == 1 && !redo && !pl.pipe && pl.rpl == 2 && pl.cj == 4 && pl.nw == 8 && pl.minw == 4 &&
                pl.mix_nfull == 0 && g_fused.fast) {
                launched = true;
                hipLaunchKernelGGL((chambolle_fused_kernel<4, 8, 4, true, false, true>), fgrid, dim3(64 * 8), 0, ctx->stream, g, pl.pbuf,
                                   pl.ctrl, pl.partials, pl.M, pl.N, pl.batch, pl.ftiles_i, pl.fnblk, steps, redo, f_out, write_f,
                                   pl.counters, kflags, sj, pl.order, FusedMix{0, 0, 0, fused_stagger(pl.fnblk)});
            } else
#ifdef SBTV_LAB
            if (!pl.pipe && pl.rpl == 2 && pl.mix_nfull > 0) {
                // mixed tiling (shipped geometry only, see prox_plan): the 128-row grid has mix_nfi tile rows
                launched = true;
                const FusedMix mix{pl.mix_nfull, pl.mix_nhi, pl.mix_row0, 0};
                if (g_fused.fast)
                    hipLaunchKernelGGL((chambolle_fused_kernel<4, 8, 4, true, true>), fgrid, dim3(64 * 8), 0, ctx->stream, g, pl.pbuf,
                                       pl.ctrl, pl.partials, pl.M, pl.N, pl.batch, pl.mix_nfi, pl.fnblk, steps, redo, f_out, write_f,
                                       pl.counters, kflags, sj, pl.order, mix);
                else
                    hipLaunchKernelGGL((chambolle_fused_kernel<4, 8, 4, false, true>), fgrid, dim3(64 * 8), 0, ctx->stream, g, pl.pbuf,
                                       pl.ctrl, pl.partials, pl.M, pl.N, pl.batch, pl.mix_nfi, pl.fnblk, steps, redo, f_out, write_f,
                                       pl.counters, kflags, sj, pl.order, mix);
            } else
#endif
            SBTV_FUSED_CASE(4, 8, 4)
#ifdef SBTV_LAB
            SBTV_FUSED_CASE(8, 4, 2)
            SBTV_FUSED_CASE(8, 8, 2)
            SBTV_FUSED_CASE(8, 8, 1)
            SBTV_FUSED_CASE(8, 4, 3)
            SBTV_FUSED_CASE(6, 8, 3)
            SBTV_FUSED_CASE(12, 4, 2)
            SBTV_FUSED_CASE(16, 4, 1)
            SBTV_FUSED_CASE(4, 8, 2)
            SBTV_FUSED_CASE(4, 16, 2)
            SBTV_FUSED_CASE(8, 6, 2)
            SBTV_FUSED_CASE(4, 8, 3)
            SBTV_FUSED_CASE(6, 8, 4)
            SBTV_FUSED_CASE(6, 8, 2)
            SBTV_FUSED_CASE(5, 8, 4)
#endif
#undef SBTV_FUSED_CASE
#define SBTV_FUSED1_CASE(CJ_, NW_, MW_)                                                                              \
    if (!pl.pipe && pl.rpl == 1 && pl.cj == CJ_ && pl.nw == NW_ && pl.minw == MW_) {             \
        launched = true;                                                                                             \
        if (g_fused.fast)                                                                                            \
            hipLaunchKernelGGL((chambolle_fused1_kernel<CJ_, NW_, MW_, true>), fgrid, dim3(64 * NW_), 0,             \
                               ctx->stream, g, pl.pbuf, pl.ctrl, pl.partials, pl.M, pl.N, pl.batch, pl.ftiles_i,     \
                               pl.fnblk, steps, redo, f_out, write_f, pl.counters, kflags, sj);                         \
        else                                                                                                         \
            hipLaunchKernelGGL((chambolle_fused1_kernel<CJ_, NW_, MW_, false>), fgrid, dim3(64 * NW_), 0,            \
                               ctx->stream, g, pl.pbuf, pl.ctrl, pl.partials, pl.M, pl.N, pl.batch, pl.ftiles_i,     \
                               pl.fnblk, steps, redo, f_out, write_f, pl.counters, kflags, sj);                         \
    }
            if (esub_wanted && !pl.esub_off && spec == 1 && !redo && !pl.pipe && pl.rpl == 1 && pl.cj == 4 && pl.nw == 8 && pl.minw == 4 && g_fused.fast) {
                launched = true;
                hipLaunchKernelGGL((chambolle_fused1_kernel<4, 8, 4, true, true>), fgrid, dim3(64 * 8), 0, ctx->stream, g, pl.pbuf,
                                   pl.ctrl, pl.partials, pl.M, pl.N, pl.batch, pl.ftiles_i, pl.fnblk, steps, redo, f_out, write_f,
                                   pl.counters, kflags, sj);
            } else
            SBTV_FUSED1_CASE(4, 8, 4)
#ifdef SBTV_LAB
            SBTV_FUSED1_CASE(4, 8, 6)
            SBTV_FUSED1_CASE(4, 8, 5)
            SBTV_FUSED1_CASE(8, 4, 6)
            SBTV_FUSED1_CASE(8, 4, 4)
            SBTV_FUSED1_CASE(8, 8, 4)
            SBTV_FUSED1_CASE(6, 8, 6)
            SBTV_FUSED1_CASE(6, 8, 4)
#endif
#undef SBTV_FUSED1_CASE
            // a plan whose geometry matches no compiled variant must not pass silently (nothing was enqueued)
            if (!launched)
                return fail(ctx, SBTV_ERR_BADARG, "prox_iterate: no fused Chambolle kernel is compiled for this tile geometry");
            // the re-run launch books its own result (last-workgroup ticket; nothing at all when it is empty)
            if (!inl && !redo && !spec_cur)
                hipLaunchKernelGGL(chambolle_fused_ctrl_kernel, dim3(pl.batch), dim3(64 * FSMAX), 0, ctx->stream,
                                   pl.ctrl, pl.partials, pl.fnblk, steps, redo, write_f);
            return 0;
        };
        const int wf = f_out ? 1 : 0;
        if (mb_main)
            for (int l = 0; l < nl; ++l) SBTV_TRY(launch_fused(base + (l < extra ? 1 : 0), 0, (l == nl - 1) ? wf : 0));
        if (mb && mb_ctrl)
            hipLaunchKernelGGL(chambolle_mb_ctrl_kernel, dim3(pl.batch), dim3(64 * FSMAX), 0, ctx->stream, pl.ctrl, pl.partials,
                               pl.fnblk, maxiter, base, extra);
        if (mb && mb_redo) SBTV_TRY(launch_fused(0, 1, wf));     // re-runs the steps up to an early stop (normally empty)
        if (!spec_cur) SBTV_TRY(launch_fused(0, 1, wf));     // redo pass; doubles as the finish-only pass when f is not valid yet
        SBTV_HIP(ctx, hipGetLastError());
        return 0;
    }
    if (cold) SBTV_TRY(prox_zero_duals(ctx, pl));     // the one-iteration kernels read the duals from memory
    for (int it = 0; it < maxiter; ++it) {
        if (v)
            hipLaunchKernelGGL(chambolle_iter_kernel<true>, grid, dim3(TVB), 0, ctx->stream, g, pl.pbuf, pl.ctrl,
                               pl.partials, pl.M, pl.N, pl.batch, pl.tiles_i);
        else
            hipLaunchKernelGGL(chambolle_iter_kernel<false>, grid, dim3(TVB), 0, ctx->stream, g, pl.pbuf, pl.ctrl,
                               pl.partials, pl.M, pl.N, pl.batch, pl.tiles_i);
        hipLaunchKernelGGL(chambolle_ctrl_kernel, dim3(pl.batch), dim3(256), 0, ctx->stream, pl.ctrl, pl.partials,
                           pl.nblk);
    }
    SBTV_HIP(ctx, hipGetLastError());
    if (f_out) return prox_finish(ctx, pl, g, f_out);
    return 0;
}

int prox_finish(sbtv_ctx *ctx, const ProxPlan &pl, const double *g, double *f) {
    const dim3 grid(pl.tiles_i, pl.tiles_j, pl.batch);
    const bool v = vec_ok(g, pl.M) && vec_ok(pl.pbuf, pl.M) && vec_ok(f, pl.M);
    if (v)
        hipLaunchKernelGGL(chambolle_finish_kernel<true>, grid, dim3(TVB), 0, ctx->stream, g, pl.pbuf, pl.ctrl, f,
                           pl.M, pl.N, pl.batch);
    else
        hipLaunchKernelGGL(chambolle_finish_kernel<false>, grid, dim3(TVB), 0, ctx->stream, g, pl.pbuf, pl.ctrl, f,
                           pl.M, pl.N, pl.batch);
    SBTV_HIP(ctx, hipGetLastError());
    return 0;
}

int reduce_jobs(sbtv_ctx *ctx, const RedJobs &jb) {
    if (jb.dropped) return fail(ctx, SBTV_ERR_BADARG, "reduce_jobs: more than four jobs in one launch (a reduction would be skipped)");
    int tot = 0;
    for (int q = 0; q < 4; ++q) tot += jb.nvec[q];
    if (tot <= 0) return 0;
    hipLaunchKernelGGL(reduce_jobs_kernel, dim3(tot), dim3(256), 0, ctx->stream, jb);
    SBTV_HIP(ctx, hipGetLastError());
    return 0;
}

int reduce_partials(sbtv_ctx *ctx, const double *partials, int nvec, int n, double *out) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(nvec), dim3(256), 0, ctx->stream, partials, n, out);
    SBTV_HIP(ctx, hipGetLastError());
    return 0;
}

int tvnorm_partials(sbtv_ctx *ctx, const double *x, int M, int N, int batch, double **partials_out, int *nblk_out) {
    const int tiles_i = (M + TI - 1) / TI, tiles_j = (N + TJ - 1) / TJ;
    const int nblk = tiles_i * tiles_j;
    double *partials = nullptr;
    SBTV_TRY(ws_get_t(ctx, "tv.partials", (size_t)batch * nblk, &partials));
    const dim3 grid(tiles_i, tiles_j, batch);
    if (vec_ok(x, M))
        hipLaunchKernelGGL(tvnorm_kernel<true>, grid, dim3(TVB), 0, ctx->stream, x, partials, M, N, tiles_i);
    else
        hipLaunchKernelGGL(tvnorm_kernel<false>, grid, dim3(TVB), 0, ctx->stream, x, partials, M, N, tiles_i);
    SBTV_HIP(ctx, hipGetLastError());
    *partials_out = partials;
    *nblk_out = nblk;
    return 0;
}

int tvnorm_dev(sbtv_ctx *ctx, const double *x, int M, int N, int batch, double *out_dev) {
    double *partials = nullptr;
    int nblk = 0;
    SBTV_TRY(tvnorm_partials(ctx, x, M, N, batch, &partials, &nblk));
    return reduce_partials(ctx, partials, batch, nblk, out_dev);
}

}  // namespace sbtv

using namespace sbtv;

extern "C" {

int sbtv_chambolle_prox_TV_stop(sbtv_ctx *ctx, const double *g, int M, int N, int batch, const double *lambda,
                                int maxiter, double tol, double tau, int warm_start, double *px, double *py,
                                double *f, int *k_out, double *err_out, int flags) {
    if (!ctx) return SBTV_ERR_BADARG;
    if (!g || !lambda || M < 2 || N < 2 || batch < 1)
        return fail(ctx, SBTV_ERR_BADARG, "chambolle_prox_TV_stop: Wrong number of required parameters");
    if (maxiter <= 0)
        return fail(ctx, SBTV_ERR_MAXITER,
                    "chambolle_prox_TV_stop: 'maxiter' is required (MaxIter undefined, chambolle_prox_TV_stop.m:131)");
    if (warm_start && (!px || !py))
        return fail(ctx, SBTV_ERR_DUALVARS, "chambolle_prox_TV_stop: Wrong size of the dual variables");
    SBTV_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cnt = (size_t)M * N * batch;
    ProxPlan pl;
    SBTV_TRY(prox_plan(ctx, M, N, batch, &pl));
    const double *gd = nullptr;
    SBTV_TRY(stage_in(ctx, "prox.in.g", g, cnt, flags, &gd));
    double *lam_d = nullptr;
    SBTV_TRY(ws_get_t(ctx, "prox.lambda", (size_t)batch, &lam_d));
    SBTV_HIP(ctx, hipMemcpyAsync(lam_d, lambda, sizeof(double) * batch, hipMemcpyHostToDevice, ctx->stream));
    SBTV_TRY(prox_reset(ctx, pl, lam_d, 1.0, maxiter, tol, tau, false, nullptr));
    if (warm_start) {
        const double *pxd = nullptr, *pyd = nullptr;
        SBTV_TRY(stage_in(ctx, "prox.in.px", px, cnt, flags, &pxd));
        SBTV_TRY(stage_in(ctx, "prox.in.py", py, cnt, flags, &pyd));
        SBTV_TRY(prox_set_duals(ctx, pl, pxd, pyd));
    }
    double *fd = nullptr, *pxo = nullptr, *pyo = nullptr;
    if (f) SBTV_TRY(stage_out_buf(ctx, "prox.out.f", f, cnt, flags, &fd));
    SBTV_TRY(prox_iterate(ctx, pl, gd, maxiter, fd, !warm_start));     // cold start: px = py = 0 (:68-69)
    if (f) SBTV_TRY(stage_out_copy(ctx, f, fd, cnt, flags));
    if (px && py) {
        SBTV_TRY(stage_out_buf(ctx, "prox.out.px", px, cnt, flags, &pxo));
        SBTV_TRY(stage_out_buf(ctx, "prox.out.py", py, cnt, flags, &pyo));
        SBTV_TRY(prox_get_duals(ctx, pl, pxo, pyo));
        SBTV_TRY(stage_out_copy(ctx, px, pxo, cnt, flags));
        SBTV_TRY(stage_out_copy(ctx, py, pyo, cnt, flags));
    }
    if (k_out || err_out || !(flags & SBTV_DEVICE_PTRS)) {
        std::vector<ProxCtrl> hc(batch);
        SBTV_HIP(ctx, hipMemcpyAsync(hc.data(), pl.ctrl, sizeof(ProxCtrl) * batch, hipMemcpyDeviceToHost, ctx->stream));
        SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int b = 0; b < batch; ++b) {
            if (k_out) k_out[b] = hc[b].k;
            if (err_out) err_out[b] = hc[b].err;
        }
    }
    return canary_epilogue(ctx, 0);
}

int sbtv_diag_prox_variant(sbtv_ctx *ctx, int M, int N, int batch, int out[6]) {
    if (!ctx || !out || M < 2 || N < 2 || batch < 1) return SBTV_ERR_BADARG;
    SBTV_HIP(ctx, hipSetDevice(ctx->device));
    ProxPlan pl;
    SBTV_TRY(prox_plan(ctx, M, N, batch, &pl));
    static const bool env_single = (getenv("SBTV_SINGLE_STEP") != nullptr);
    out[0] = pl.cj;
    out[1] = pl.nw;
    out[2] = pl.minw;
    out[3] = pl.rpl;
    out[4] = pl.fnblk;
    out[5] = (M % 2 == 0 && !env_single && !g_force_single_step) ? (pl.pipe ? 2 : 1) : 0;
    return 0;
}

int sbtv_TVnorm(sbtv_ctx *ctx, const double *x, int M, int N, int batch, double *out, int flags) {
    if (!ctx) return SBTV_ERR_BADARG;
    if (!x || !out || M < 2 || N < 2 || batch < 1) return fail(ctx, SBTV_ERR_BADARG, "TVnorm: bad arguments");
    SBTV_HIP(ctx, hipSetDevice(ctx->device));
    const double *xd = nullptr;
    SBTV_TRY(stage_in(ctx, "tv.in.x", x, (size_t)M * N * batch, flags, &xd));
    double *od = nullptr;
    SBTV_TRY(ws_get_t(ctx, "tv.out", (size_t)batch, &od));
    SBTV_TRY(tvnorm_dev(ctx, xd, M, N, batch, od));
    SBTV_HIP(ctx, hipMemcpyAsync(out, od, sizeof(double) * batch, hipMemcpyDeviceToHost, ctx->stream));
    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return canary_epilogue(ctx, 0);
}

}  // extern "C"

#ifdef SBTV_FUSED_TIMELINE
extern "C" int sbtv_debug_timeline(sbtv_ctx *ctx, unsigned long long *out, int nrec) {
    if (!ctx || !out || nrec < 1 || nrec > 8192) return SBTV_ERR_BADARG;
    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SBTV_HIP(ctx, hipMemcpyFromSymbol(out, HIP_SYMBOL(sbtv::g_fused_tl), sizeof(unsigned long long) * 8 * (size_t)nrec));
    return 0;
}
// per-wave barrier stamps of the last launch: out[TLW_SAMPLES][16 waves][TLW_SLOTS]
extern "C" int sbtv_debug_timeline_waves(sbtv_ctx *ctx, unsigned long long *out, int *samples, int *every, int *slots) {
    if (!ctx || !out) return SBTV_ERR_BADARG;
    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SBTV_HIP(ctx, hipMemcpyFromSymbol(out, HIP_SYMBOL(sbtv::g_fused_tlw), sizeof(sbtv::g_fused_tlw)));
    if (samples) *samples = sbtv::TLW_SAMPLES;
    if (every) *every = sbtv::TLW_EVERY;
    if (slots) *slots = sbtv::TLW_SLOTS;
    return 0;
}
#endif
