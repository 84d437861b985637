// Batched 2-D real FFT / IFFT and the spectral form of the circular blur
// operator (K5-K8), hand-written for gfx950.  No rocFFT.
//
// A real M x N image (column-major) is transformed in two passes:
//   cols : each column (M reals = M/2 complex, contiguous) -> complex FFT of
//          length n1 = M/2 + the real-input split step.  The result is the
//          PACKED half spectrum S (n1 x N complex, column-major): row k>=1 is
//          X[k,:], row 0 holds X[0,:] + i X[M/2,:] (both are real sequences
//          along the column, so nothing is lost and S has exactly the bytes of
//          the real image).
//   rows : each of the n1 spectral rows (stride n1) -> complex FFT of length N.
//          A workgroup owns RK consecutive rows so a wave reads RK*16-byte
//          contiguous segments.  The rows kernel optionally applies a spectral
//          point-wise operator and the inverse row FFT before storing, so that
//          A, AT, invLS and the fused SALSA / gradient steps need one pass.
// The 1-D engine is a mixed-radix (8,8,..,{4,2}) Stockham autosort FFT: every
// thread keeps 8 complex values in registers (elements t + s*n/8), stages are
// radix-8 butterflies in registers and the inter-stage exchange goes through
// LDS (real and imaginary planes in turn, padded against bank conflicts).
// The first stage reads global memory directly in its natural coalesced order
// and the last stage leaves the result in the same register slots, so global
// stores are coalesced too.
#include <type_traits>

#include "sbtv_internal.h"

namespace sbtv {

// ---------------------------------------------------------------------------
// complex helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cmulc(double2 a, double2 b) {   // a * conj(b)
    return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ double2 cconj(double2 a) { return make_double2(a.x, -a.y); }
__device__ __forceinline__ double2 cscale(double2 a, double s) { return make_double2(a.x * s, a.y * s); }
__device__ __forceinline__ double cabs2(double2 a) { return a.x * a.x + a.y * a.y; }

// Tiled packed-spectrum layout of the sizes that take the wave-granular column kernels (fft_wave.inc):
// S[(l / TW)][k][l % TW], i.e. 64-byte pieces of 4 columns of one row k.
constexpr int TWS = 2;                    // log2 of the tile width
constexpr int TW = 1 << TWS;
__device__ __forceinline__ size_t s_tiled(int k, int l, int n1) {
    return (((size_t)(l >> TWS) * n1 + k) << TWS) + (l & (TW - 1));
}
// operator spectra in the same tiling (n1 + 1 rows: k = 0..M/2): a workgroup's 4 rows x 4 columns are 256 contiguous bytes
__device__ __forceinline__ size_t u_tiled(int k, int l, int n1) { return ((size_t)(l >> TWS) * (n1 + 1) + k) * TW + (l & (TW - 1)); }

template <bool INV>
__device__ __forceinline__ double2 twid(const double2 *__restrict__ tw, int idx) {
    double2 w = tw[idx];
    if (INV) w.y = -w.y;
    return w;
}
// a twiddle table kept in LDS, one pad element per 16: the stage twiddles are read at power-of-two strides
// ((r k) << SH), which would put the lanes of a group on the same banks of an unpadded table
struct LdsTw {
    const double2 *p;
    static __host__ __device__ constexpr int elems(int n) { return n + (n >> 4); }
    static __device__ __forceinline__ int idx(int i) { return i + (i >> 4); }
};
template <bool INV>
__device__ __forceinline__ double2 twid(LdsTw tw, int idx) {
    double2 w = tw.p[LdsTw::idx(idx)];
    if (INV) w.y = -w.y;
    return w;
}
// multiply by -i (forward) / +i (inverse)
template <bool INV>
__device__ __forceinline__ double2 mul_mi(double2 a) {
    return INV ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x);
}

template <bool INV>
__device__ __forceinline__ void bfly2(double2 &a, double2 &b) {
    const double2 t = a;
    a = cadd(t, b);
    b = csub(t, b);
}
template <bool INV>
__device__ __forceinline__ void bfly4(double2 &a0, double2 &a1, double2 &a2, double2 &a3) {
    const double2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = mul_mi<INV>(csub(a1, a3));
    a0 = cadd(t0, t2);
    a1 = cadd(t1, t3);
    a2 = csub(t0, t2);
    a3 = csub(t1, t3);
}
template <bool INV>
__device__ __forceinline__ void bfly8(double2 &a0, double2 &a1, double2 &a2, double2 &a3, double2 &a4, double2 &a5,
                                      double2 &a6, double2 &a7) {
    constexpr double R2 = 0.70710678118654752440;
    bfly4<INV>(a0, a2, a4, a6);   // E[0..3] in a0,a2,a4,a6
    bfly4<INV>(a1, a3, a5, a7);   // O[0..3] in a1,a3,a5,a7
    // w8^m * O[m]
    double2 o1, o3;
    if (!INV) {
        o1 = make_double2((a3.x + a3.y) * R2, (a3.y - a3.x) * R2);      // (1-i)/sqrt2
        o3 = make_double2((a7.y - a7.x) * R2, (-a7.x - a7.y) * R2);     // (-1-i)/sqrt2
    } else {
        o1 = make_double2((a3.x - a3.y) * R2, (a3.x + a3.y) * R2);      // (1+i)/sqrt2
        o3 = make_double2((-a7.x - a7.y) * R2, (a7.x - a7.y) * R2);     // (-1+i)/sqrt2
    }
    const double2 o0 = a1, o2 = mul_mi<INV>(a5);
    const double2 e0 = a0, e1 = a2, e2 = a4, e3 = a6;
    a0 = cadd(e0, o0);
    a1 = cadd(e1, o1);
    a2 = cadd(e2, o2);
    a3 = cadd(e3, o3);
    a4 = csub(e0, o0);
    a5 = csub(e1, o1);
    a6 = csub(e2, o2);
    a7 = csub(e3, o3);
}

template <int LR, bool INV>
__device__ __forceinline__ void butterflies(double2 (&v)[8]) {
    if constexpr (LR == 3) {
        bfly8<INV>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    } else if constexpr (LR == 2) {
        bfly4<INV>(v[0], v[2], v[4], v[6]);
        bfly4<INV>(v[1], v[3], v[5], v[7]);
    } else {
        bfly2<INV>(v[0], v[4]);
        bfly2<INV>(v[1], v[5]);
        bfly2<INV>(v[2], v[6]);
        bfly2<INV>(v[3], v[7]);
    }
}

// twiddles of a Stockham stage: element r of the butterfly at j gets
// exp(-+2 pi i r (j mod Ns) / (Ns R))
template <int LOG2N, int LOG2NS, int LR, bool INV>
__device__ __forceinline__ void apply_twiddles(double2 (&v)[8], int t, const double2 *__restrict__ tw) {
    if constexpr (LOG2NS > 0) {
        constexpr int T = (1 << LOG2N) / 8;
        constexpr int R = 1 << LR, G = 8 / R;
        constexpr int SH = LOG2N - LOG2NS - LR;   // log2 of n / (Ns R)
#pragma unroll
        for (int q = 0; q < G; ++q) {
            const int k = (t + q * T) & ((1 << LOG2NS) - 1);
#pragma unroll
            for (int r = 1; r < R; ++r) v[q + r * G] = cmul(v[q + r * G], twid<INV>(tw, (r * k) << SH));
        }
    }
}

// LDS exchange accessors ------------------------------------------------------
// Column kernels: each sequence has its own padded region, t is the fast lane index.
struct ColsX {
    double *lds;     // base of this sequence's region (real plane)
    int im_off;      // offset (in doubles) of the imaginary plane; 0 = one plane used for re and im in turn
    __device__ __forceinline__ int idx(int pos) const { return pos + (pos >> 3); }
};
// Row kernels: RK sequences interleaved (sequence index fastest).
template <int RK>
struct RowsX {
    double *lds;
    int im_off;
    int q;
    __device__ __forceinline__ int idx(int pos) const {
        const int i = pos * RK + q;
        return i + ((i >> 5) << 2);
    }
};

// write stage outputs to their Stockham positions, read back elements t + s*T
template <int LOG2N, int LOG2NS, int LR, class X>
__device__ __forceinline__ void exchange(double2 (&v)[8], int t, const X &x) {
    constexpr int T = (1 << LOG2N) / 8;
    constexpr int R = 1 << LR, G = 8 / R;
    int pos[8];
#pragma unroll
    for (int q = 0; q < G; ++q) {
        const int j = t + q * T;
        const int k = j & ((1 << LOG2NS) - 1);
        const int j0 = ((j >> LOG2NS) << (LOG2NS + LR)) + k;
#pragma unroll
        for (int r = 0; r < R; ++r) pos[q + r * G] = x.idx(j0 + (r << LOG2NS));
    }
    if (x.im_off) {
        // both planes resident: one write phase, one read phase (2 barriers)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            x.lds[pos[s]] = v[s].x;
            x.lds[pos[s] + x.im_off] = v[s].y;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int i = x.idx(t + s * T);
            v[s].x = x.lds[i];
            v[s].y = x.lds[i + x.im_off];
        }
        __syncthreads();
        return;
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) x.lds[pos[s]] = v[s].x;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) v[s].x = x.lds[x.idx(t + s * T)];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) x.lds[pos[s]] = v[s].y;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) v[s].y = x.lds[x.idx(t + s * T)];
    __syncthreads();
}

// mirrored read: out[s] = in[(n - (t+s*T)) mod n]
template <int LOG2N, class X>
__device__ __forceinline__ void mirror(const double2 (&v)[8], double2 (&m)[8], int t, const X &x) {
    constexpr int n = 1 << LOG2N, T = n / 8;
    if (x.im_off) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int i = x.idx(t + s * T);
            x.lds[i] = v[s].x;
            x.lds[i + x.im_off] = v[s].y;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int i = x.idx((n - (t + s * T)) & (n - 1));
            m[s].x = x.lds[i];
            m[s].y = x.lds[i + x.im_off];
        }
        __syncthreads();
        return;
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) x.lds[x.idx(t + s * T)] = v[s].x;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) m[s].x = x.lds[x.idx((n - (t + s * T)) & (n - 1))];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) x.lds[x.idx(t + s * T)] = v[s].y;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) m[s].y = x.lds[x.idx((n - (t + s * T)) & (n - 1))];
    __syncthreads();
}

template <int LOG2N, int LOG2NS, bool INV, class X>
__device__ __forceinline__ void fft_stages(double2 (&v)[8], int t, const double2 *__restrict__ tw, const X &x) {
    constexpr int left = LOG2N - LOG2NS;
    if constexpr (left > 0) {
        constexpr int LR = left >= 3 ? 3 : left;
        apply_twiddles<LOG2N, LOG2NS, LR, INV>(v, t, tw);
        butterflies<LR, INV>(v);
        if constexpr (left - LR > 0) {
            exchange<LOG2N, LOG2NS, LR>(v, t, x);
            fft_stages<LOG2N, LOG2NS + LR, INV>(v, t, tw, x);
        }
    }
}

// ---------------------------------------------------------------------------
// column pass, forward: real columns -> packed half spectrum
// ---------------------------------------------------------------------------
constexpr int COLS_THREADS = 256;

template <int LOG2N>   // n1 = M/2 = 1 << LOG2N
__global__ __launch_bounds__(COLS_THREADS) void fft_cols_fwd_kernel(const double *__restrict__ x,
                                                                     const double *__restrict__ add,
                                                                     double2 *__restrict__ S,
                                                                     const double2 *__restrict__ tw_n1,
                                                                     const double2 *__restrict__ tw_M, int N,
                                                                     const int *__restrict__ frozen,
                                                                     double *__restrict__ tvp) {
    constexpr int n = 1 << LOG2N, T = n / 8;
    constexpr int LDSN = n + (n >> 3);
    constexpr int MAXSEQ = (COLS_THREADS / T) > 0 ? (COLS_THREADS / T) : 1;
    __shared__ double lds[2 * MAXSEQ * LDSN];
    const int b = blockIdx.y;
    if (frozen && frozen[b]) return;
    const int nseq = blockDim.x / T;
    const int seq = threadIdx.x / T, t = threadIdx.x - seq * T;
    const int j = blockIdx.x * nseq + seq;          // column (grid is exact: N % nseq == 0)
    const size_t colbase = ((size_t)b * N + j) * n;  // in complex elements
    const double2 *__restrict__ xin = reinterpret_cast<const double2 *>(x) + colbase;
    ColsX X{lds + seq * LDSN, MAXSEQ * LDSN};
    double2 v[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) v[s] = xin[t + s * T];
    if (tvp) {
        // periodic isotropic TV of x (utils/TVnorm.m:2) on the way: tvp[b][workgroup] = this workgroup's share, summed in
        // a fixed order.  Saves the callers (SAPG, FISTA) a separate TV launch over the same image.
        constexpr int M = 2 * n;
        const size_t cb = ((size_t)b * N + j) * M, cl = ((size_t)b * N + (j > 0 ? j - 1 : N - 1)) * M;
        double up[8];
        double2 ul[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int e = t + s * T;
            up[s] = x[e > 0 ? cb + 2 * (size_t)e - 1 : cb + M - 1];
            ul[s] = *reinterpret_cast<const double2 *>(x + cl + 2 * (size_t)e);
        }
        double a = 0.0;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const double h0 = v[s].x - ul[s].x, v0 = v[s].x - up[s], h1 = v[s].y - ul[s].y, v1 = v[s].y - v[s].x;
            a += sqrt(h0 * h0 + v0 * v0) + sqrt(h1 * h1 + v1 * v1);
        }
        const int nthr = blockDim.x;
        lds[threadIdx.x] = a;
        __syncthreads();
        for (int stride = nthr >> 1; stride > 0; stride >>= 1) {
            if ((int)threadIdx.x < stride) lds[threadIdx.x] += lds[threadIdx.x + stride];
            __syncthreads();
        }
        if (threadIdx.x == 0) tvp[(size_t)b * gridDim.x + blockIdx.x] = lds[0];
        __syncthreads();
    }
    if (add) {
        const double2 *__restrict__ ain = reinterpret_cast<const double2 *>(add) + colbase;
#pragma unroll
        for (int s = 0; s < 8; ++s) v[s] = cadd(v[s], ain[t + s * T]);
    }
    fft_stages<LOG2N, 0, false>(v, t, tw_n1, X);
    double2 m[8];
    mirror<LOG2N>(v, m, t, X);
    double2 *__restrict__ out = S + colbase;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int e = t + s * T;
        const double2 Z = v[s], Zm = cconj(m[s]);
        double2 r;
        if (e == 0) {
            r = make_double2(Z.x + Z.y, Z.x - Z.y);          // X[0] + i X[M/2]
        } else {
            const double2 w = tw_M[e];
            const double2 su = cadd(Z, Zm), di = csub(Z, Zm);
            const double2 wd = cmul(w, di);                  // w (Z - Zm)
            // X = 1/2 su - i/2 wd
            r = make_double2(0.5 * (su.x + wd.y), 0.5 * (su.y - wd.x));
        }
        out[e] = r;
    }
}

// column pass, inverse: packed half spectrum -> real columns (times `scale`).
// POST = true additionally runs the SALSA bookkeeping pass on the column while x is still in
// registers (SALSA_v2.m:440-451): bu += u - x ; g = x - bu ; partial sums of (x-true)^2, (x-u)^2,
// x^2, u^2, (x-xprev)^2 and the periodic TV of u -> post.partials[b][6][gridDim.x].
// Non-temporal 16-byte accesses for the two streams of the fused bookkeeping pass that nothing reads again in the
// same outer iteration (the x store, the `true` load): they then do not evict lines that are re-used within
// microseconds (duals between the two Chambolle launches, g, u, S).  Measured +0.5 % SALSA it/s at 2048^2; the same
// hint on the H / Y loads of the row pass costs 4 % (those do profit from the Infinity Cache across iterations).
typedef double sbtv_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 nt_load2(const double *p) {
    const sbtv_d2 v = __builtin_nontemporal_load(reinterpret_cast<const sbtv_d2 *>(p));
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ void nt_store2(double *p, double2 v) {
    sbtv_d2 w;
    w.x = v.x;
    w.y = v.y;
    __builtin_nontemporal_store(w, reinterpret_cast<sbtv_d2 *>(p));
}
template <int LOG2N, bool POST>
__global__ __launch_bounds__(COLS_THREADS) void fft_cols_inv_kernel(const double2 *__restrict__ S,
                                                                     double *__restrict__ x,
                                                                     const double2 *__restrict__ tw_n1,
                                                                     const double2 *__restrict__ tw_M, int N,
                                                                     double scale, const int *__restrict__ frozen,
                                                                     ColsPost post) {
    constexpr int n = 1 << LOG2N, T = n / 8;
    constexpr int LDSN = n + (n >> 3);
    constexpr int MAXSEQ = (COLS_THREADS / T) > 0 ? (COLS_THREADS / T) : 1;
    __shared__ double lds[2 * MAXSEQ * LDSN];
    const int b = blockIdx.y;
    if (frozen && frozen[b]) return;
    const int nseq = blockDim.x / T;
    const int seq = threadIdx.x / T, t = threadIdx.x - seq * T;
    const int j = blockIdx.x * nseq + seq;
    const size_t colbase = ((size_t)b * N + j) * n;
    const double2 *__restrict__ in = S + colbase;
    ColsX X{lds + seq * LDSN, MAXSEQ * LDSN};
    double2 v[8], m[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) v[s] = in[t + s * T];
    // Small images (M <= 512: at most 256 workgroups of 64 threads, one per CU): the pass is a chain of memory round
    // trips, so the bookkeeping streams of the epilogue are requested here, together with the spectrum column, instead
    // of element by element after the transform (the compiler waits for every element's loads separately there).
    constexpr bool PREP = POST && (LOG2N <= 8);
    double2 pu[PREP ? 8 : 1], pb[PREP ? 8 : 1], pt[PREP ? 8 : 1], pul[PREP ? 8 : 1], pxo[PREP ? 8 : 1];
    double pup[PREP ? 8 : 1];
    if constexpr (PREP) {
        constexpr int M = 2 * n;
        const size_t ibase = (size_t)b * N * M, cb = ibase + (size_t)j * M;
        const size_t cl = ibase + (size_t)(j > 0 ? j - 1 : N - 1) * M;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int e = t + s * T;
            const size_t o = cb + 2 * (size_t)e;
            pu[s] = *reinterpret_cast<const double2 *>(post.u + o);
            pb[s] = *reinterpret_cast<const double2 *>(post.bu + o);
            if (post.tru) pt[s] = nt_load2(post.tru + o);
            if (post.xprev) pxo[s] = *reinterpret_cast<const double2 *>(post.xprev + o);
            pup[s] = post.u[e > 0 ? o - 1 : cb + M - 1];
            pul[s] = *reinterpret_cast<const double2 *>(post.u + cl + 2 * (size_t)e);
        }
    }
    mirror<LOG2N>(v, m, t, X);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int e = t + s * T;
        if (e == 0) {
            const double X0 = v[s].x, Xn = v[s].y;
            v[s] = make_double2(0.5 * (X0 + Xn), 0.5 * (X0 - Xn));     // E + i O
        } else {
            const double2 Xk = v[s], Xm = cconj(m[s]);
            const double2 E = cscale(cadd(Xk, Xm), 0.5);
            const double2 O = cmulc(cscale(csub(Xk, Xm), 0.5), tw_M[e]);   // * conj(w_M^e)
            v[s] = make_double2(E.x - O.y, E.y + O.x);                  // E + i O
        }
    }
    fft_stages<LOG2N, 0, true>(v, t, tw_n1, X);
    double2 *__restrict__ out = reinterpret_cast<double2 *>(x) + colbase;
    if constexpr (!POST) {
#pragma unroll
        for (int s = 0; s < 8; ++s) out[t + s * T] = cscale(v[s], scale);
    } else {
        constexpr int M = 2 * n;
        const size_t ibase = (size_t)b * N * M;               // image offset in doubles
        const size_t cb = ibase + (size_t)j * M;              // column offset
        const size_t cl = ibase + (size_t)(j > 0 ? j - 1 : N - 1) * M;   // periodic left neighbour column
        double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int e = t + s * T;
            const size_t o = cb + 2 * (size_t)e;
            const double2 xv = cscale(v[s], scale);
            nt_store2(reinterpret_cast<double *>(out + e), xv);       // x is not read again in this iteration
            const double2 uv = PREP ? pu[PREP ? s : 0] : *reinterpret_cast<const double2 *>(post.u + o);
            double2 bv = PREP ? pb[PREP ? s : 0] : *reinterpret_cast<const double2 *>(post.bu + o);
            bv.x = bv.x + (uv.x - xv.x);
            bv.y = bv.y + (uv.y - xv.y);
            *reinterpret_cast<double2 *>(post.bu + o) = bv;
            *reinterpret_cast<double2 *>(post.g + o) = make_double2(xv.x - bv.x, xv.y - bv.y);
            if (post.tru) {
                const double2 tv = PREP ? pt[PREP ? s : 0] : nt_load2(post.tru + o);
                const double e0 = xv.x - tv.x, e1 = xv.y - tv.y;
                acc[0] += e0 * e0 + e1 * e1;
            }
            const double d0 = xv.x - uv.x, d1 = xv.y - uv.y;
            acc[1] += d0 * d0 + d1 * d1;
            acc[2] += xv.x * xv.x + xv.y * xv.y;
            acc[3] += uv.x * uv.x + uv.y * uv.y;
            if (post.xprev) {
                const double2 xo = PREP ? pxo[PREP ? s : 0] : *reinterpret_cast<const double2 *>(post.xprev + o);
                const double p0 = xv.x - xo.x, p1 = xv.y - xo.y;
                acc[4] += p0 * p0 + p1 * p1;
            }
            // periodic isotropic TV of u (utils/TVnorm.m:2)
            const double up = PREP ? pup[PREP ? s : 0] : post.u[e > 0 ? o - 1 : cb + M - 1];
            const double2 ul = PREP ? pul[PREP ? s : 0] : *reinterpret_cast<const double2 *>(post.u + cl + 2 * (size_t)e);
            const double h0 = uv.x - ul.x, v0 = uv.x - up, h1 = uv.y - ul.y, v1 = uv.y - uv.x;
            acc[5] += sqrt(h0 * h0 + v0 * v0) + sqrt(h1 * h1 + v1 * v1);
        }
        // fixed-order block reduction through LDS (the exchange planes are idle now)
        __syncthreads();
        const int nthr = blockDim.x;
#pragma unroll
        for (int c = 0; c < 6; ++c) lds[c * nthr + threadIdx.x] = acc[c];
        __syncthreads();
        for (int stride = nthr >> 1; stride > 0; stride >>= 1) {
            if ((int)threadIdx.x < stride) {
#pragma unroll
                for (int c = 0; c < 6; ++c) lds[c * nthr + threadIdx.x] += lds[c * nthr + threadIdx.x + stride];
            }
            __syncthreads();
        }
        if (threadIdx.x < 6) post.partials[((size_t)b * 6 + threadIdx.x) * gridDim.x + blockIdx.x] = lds[threadIdx.x * nthr];
    }
}

// ---------------------------------------------------------------------------
// row pass with optional fused spectral operator
// ---------------------------------------------------------------------------
struct RowsParams {
    const double2 *Sin;
    double2 *Sout;
    const double2 *tw;      // length N
    const double2 *H, *Y, *D1, *D2;
    double2 *E;             // OP_CSALSA: state spectrum (layout of H), updated in place
    const double *cs;       // OP_CSALSA: coefficients (device)
    const double *mu;
    double *acc;            // [batch][3][nrb]
    const int *frozen;
    int n1;                 // rows of S per image
    int fwd, inv, op;
    int shared_spec;        // 1: H / Y / D1 / D2 hold ONE spectrum shared by every image of the batch
    int fold;               // pipelined row kernel: > 0 = the batch folded into grid.x (see the kernel), else 0
    size_t u_img;           // elements of one operator spectrum per image
    int u_ld;               // leading dimension of the row-major operator spectra (tiled mode)
    int u_tiled;            // operator spectra in the tiled layout U[(l/4)][k][l%4], k = 0..n1
};

template <int OP>
__device__ __forceinline__ double2 spec_apply(double2 V, double2 H, double2 Y, double2 D1, double2 D2, double mu,
                                              double wgt, double (&acc)[3]) {
    if constexpr (OP == OP_MUL_H) {
        return cmul(V, H);
    } else if constexpr (OP == OP_MUL_HC) {
        return cmulc(V, H);
    } else if constexpr (OP == OP_INVLS) {
        const double d = cabs2(H) + mu;
        return make_double2(V.x / d, V.y / d);
    } else if constexpr (OP == OP_ATA) {
        return cscale(V, cabs2(H));
    } else if constexpr (OP == OP_SALSA) {
        const double d = cabs2(H) + mu;
        const double2 num = cadd(cmulc(Y, H), cscale(V, mu));     // conj(H) Y + mu S
#ifdef SBTV_SPEC_IEEE_DIV
        const double rd = 1.0 / d;                                // one IEEE reciprocal instead of two divisions
#else
        const double rd = fast_rcp(d);                            // d in [mu, 1 + mu]: rcp + one cubic step (~1 ulp)
#endif
        const double2 Xh = make_double2(num.x * rd, num.y * rd);
        const double2 R = csub(Y, cmul(H, Xh));
        acc[0] += wgt * cabs2(R);
        return Xh;
    } else if constexpr (OP == OP_RESID) {
        const double2 R = csub(cmul(H, V), Y);
        acc[0] += wgt * cabs2(R);
        return V;
    } else if constexpr (OP == OP_GRADF) {
        const double2 R = csub(cmul(H, V), Y);
        acc[0] += wgt * cabs2(R);
        return cmulc(R, H);
    } else if constexpr (OP == OP_GRAD) {
        const double2 R = csub(cmul(H, V), Y);
        acc[0] += wgt * cabs2(R);
        const double2 d1 = cmulc(cmul(D1, V), R), d2 = cmulc(cmul(D2, V), R);
        acc[1] += wgt * d1.x;
        acc[2] += wgt * d2.x;
        return cmulc(R, H);                                       // conj(H) R
    } else {
        return V;
    }
}

// OP_CSALSA (see SpecOp): E is this element of the state spectrum, replaced by its new value
struct CsPar {
    double c_y, c_e, c_keep;
};
__device__ __forceinline__ double2 spec_apply_cs(double2 V, double2 H, double2 Y, double2 &E, double mu, const CsPar &c,
                                                 double wgt, double (&acc)[3]) {
    const double d = cabs2(H) + mu;
    const double2 W = cadd(cscale(Y, c.c_y), cscale(E, c.c_e));
    const double2 num = cadd(cmulc(W, H), cscale(V, mu));         // conj(H) W + mu S
#ifdef SBTV_SPEC_IEEE_DIV
    const double rd = 1.0 / d;
#else
    const double rd = fast_rcp(d);                                // d in [mu, 1 + mu]
#endif
    const double2 Xh = make_double2(num.x * rd, num.y * rd);
    const double2 T = csub(cmul(H, Xh), Y);                       // spectrum of A x - y
    const double2 En = cadd(T, cscale(E, c.c_keep));
    acc[0] += wgt * cabs2(T);
    acc[1] += wgt * cabs2(En);
    acc[2] += wgt * cabs2(csub(En, E));
    E = En;
    return Xh;
}

// TILED: S is in the tiled layout (a workgroup's RK rows x 4 columns are then 64 RK contiguous bytes) and the operator
// spectra are row-major U[k][l] (fft_wave.inc); otherwise S[l][k] and U[l][k].
template <int LOG2N, int RK, int OP, bool TILED = false>
__global__ __launch_bounds__(RK *((1 << LOG2N) / 8), (TILED && RK == 2) ? 4 : 1) void fft_rows_kernel(RowsParams p) {
    constexpr int N = 1 << LOG2N, T = N / 8;
    constexpr int LDSI = RK * N;
    constexpr int LDSN = LDSI + ((LDSI >> 5) << 2) + 8;
    __shared__ double lds[2 * LDSN];
    __shared__ double red[3 * 16];
    const int b = blockIdx.y;
    if (p.frozen && p.frozen[b]) return;
    const int q = threadIdx.x % RK, t = threadIdx.x / RK;
    // XCD-aware row-block order (blockIdx % 8 names the group of workgroups sharing one L2): each
    // group walks a contiguous range of row blocks, so the 128-byte lines that consecutive row
    // blocks share (RK * 16 B < 128 B) are re-used from that L2.  Bijective for any block count.
    int kb;
    {
        const int nt = gridDim.x, bid = blockIdx.x;
        const int q8 = nt >> 3, r8 = nt & 7, x = bid & 7, o = bid >> 3;
        kb = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + o;
    }
    const int k = kb * RK + q;
    const int n1 = p.n1;
    const size_t ibase = (size_t)b * n1 * N;
    const double2 *__restrict__ in = p.Sin + ibase;
    RowsX<RK> X{lds, LDSN, q};
    double2 v[8];
    auto sidx = [&](int l) -> size_t { return TILED ? s_tiled(k, l, n1) : (size_t)l * n1 + k; };
    auto uidx = [&](int kk, int l) -> size_t { return (TILED && p.u_tiled) ? u_tiled(kk, l, n1) : (size_t)l * (n1 + 1) + kk; };
#pragma unroll
    for (int s = 0; s < 8; ++s) v[s] = in[sidx(t + s * T)];
    // Small workgroups (<= 256 threads: images up to 512^2 and the RK = 1 / 2 variants) have registers to spare and are
    // bound by the chain of memory round trips, not by throughput: they request the operator spectra of their row
    // together with the row itself, BEFORE the forward transform (one round trip instead of nine: written as a loop
    // the compiler waits for every spectrum value separately).  The workgroup with the packed row 0 keeps the loop.
    constexpr bool CS = (OP == OP_CSALSA);       // its own branch below (the state spectrum is read AND written)
    constexpr bool PRE = (OP != OP_NONE) && !CS && (RK * T <= 256);
    constexpr bool needY = (OP == OP_SALSA || OP == OP_RESID || OP == OP_GRAD || OP == OP_GRADF || CS);
    constexpr bool needD = (OP == OP_GRAD);
    const size_t hbase = p.shared_spec ? 0 : (size_t)b * p.u_img;
    double2 hh[PRE ? 8 : 1], yy[(PRE && needY) ? 8 : 1], dd1[(PRE && needD) ? 8 : 1], dd2[(PRE && needD) ? 8 : 1];
    // spectrum row M/2, for the thread row that owns the packed row 0 (its own row k = 0 is covered by hh / yy)
    double2 hq[PRE ? 8 : 1], yq[(PRE && needY) ? 8 : 1], d1q[(PRE && needD) ? 8 : 1], d2q[(PRE && needD) ? 8 : 1];
    if constexpr (PRE) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const size_t hi = hbase + uidx(k, t + s * T);
            hh[s] = p.H[hi];
            if constexpr (needY) yy[s] = p.Y[hi];
            if constexpr (needD) {
                dd1[s] = p.D1[hi];
                dd2[s] = p.D2[hi];
            }
        }
        if (kb == 0 && q == 0) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const size_t hn = hbase + uidx(n1, t + s * T);
                hq[s] = p.H[hn];
                if constexpr (needY) yq[s] = p.Y[hn];
                if constexpr (needD) {
                    d1q[s] = p.D1[hn];
                    d2q[s] = p.D2[hn];
                }
            }
        }
    }
    if (p.fwd) fft_stages<LOG2N, 0, false>(v, t, p.tw, X);

    double acc[3] = {0.0, 0.0, 0.0};
    if constexpr (OP != OP_NONE) {
        const double mu = p.mu ? p.mu[b] : 0.0;
        if constexpr (CS) {
            const CsPar cp{p.cs[0], p.cs[1], p.cs[2]};
            double2 *__restrict__ Ep = p.E + hbase;
            const double2 *__restrict__ Hp = p.H + hbase, *__restrict__ Yp = p.Y + hbase;
            if (kb == 0) {
                double2 m[8];
                mirror<LOG2N>(v, m, t, X);
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const int l = t + s * T;
                    if (q == 0) {
                        const size_t h0 = uidx(0, l), hn = uidx(n1, l);
                        const double2 P = v[s], Q = cconj(m[s]);
                        const double2 A = cscale(cadd(P, Q), 0.5);
                        const double2 dB = csub(P, Q);
                        const double2 B = make_double2(0.5 * dB.y, -0.5 * dB.x);   // (P-Q)/(2i)
                        double2 e0 = Ep[h0], en = Ep[hn];
                        const double2 A2 = spec_apply_cs(A, Hp[h0], Yp[h0], e0, mu, cp, 1.0, acc);
                        const double2 B2 = spec_apply_cs(B, Hp[hn], Yp[hn], en, mu, cp, 1.0, acc);
                        Ep[h0] = e0;
                        Ep[hn] = en;
                        v[s] = make_double2(A2.x - B2.y, A2.y + B2.x);             // A' + i B'
                    } else {
                        const size_t hk = uidx(k, l);
                        double2 e = Ep[hk];
                        v[s] = spec_apply_cs(v[s], Hp[hk], Yp[hk], e, mu, cp, 2.0, acc);
                        Ep[hk] = e;
                    }
                }
            } else {
                double2 hh2[8], yy2[8], ee2[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const size_t hk = uidx(k, t + s * T);
                    hh2[s] = Hp[hk];
                    yy2[s] = Yp[hk];
                    ee2[s] = Ep[hk];
                }
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    v[s] = spec_apply_cs(v[s], hh2[s], yy2[s], ee2[s], mu, cp, 2.0, acc);
                    Ep[uidx(k, t + s * T)] = ee2[s];
                }
            }
        } else if (kb == 0) {
            // the block that owns the packed row 0 = X[0,:] + i X[M/2,:]
            double2 m[8];
            mirror<LOG2N>(v, m, t, X);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int l = t + s * T;
                const size_t hi = hbase + uidx(0, l);
                const size_t hin = hbase + uidx(n1, l), hik = hbase + uidx(k, l);
                const double2 z = make_double2(0.0, 0.0);
                if (q == 0) {
                    const double2 P = v[s], Q = cconj(m[s]);
                    const double2 A = cscale(cadd(P, Q), 0.5);
                    const double2 dB = csub(P, Q);
                    const double2 B = make_double2(0.5 * dB.y, -0.5 * dB.x);   // (P-Q)/(2i)
                    double2 A2, B2;
                    if constexpr (PRE) {
                        A2 = spec_apply<OP>(A, hh[s], needY ? yy[needY ? s : 0] : z, needD ? dd1[needD ? s : 0] : z,
                                            needD ? dd2[needD ? s : 0] : z, mu, 1.0, acc);
                        B2 = spec_apply<OP>(B, hq[s], needY ? yq[needY ? s : 0] : z, needD ? d1q[needD ? s : 0] : z,
                                            needD ? d2q[needD ? s : 0] : z, mu, 1.0, acc);
                    } else {
                        A2 = spec_apply<OP>(A, p.H[hi], needY ? p.Y[hi] : z, needD ? p.D1[hi] : z, needD ? p.D2[hi] : z, mu,
                                            1.0, acc);
                        B2 = spec_apply<OP>(B, p.H[hin], needY ? p.Y[hin] : z, needD ? p.D1[hin] : z,
                                            needD ? p.D2[hin] : z, mu, 1.0, acc);
                    }
                    v[s] = make_double2(A2.x - B2.y, A2.y + B2.x);             // A' + i B'
                } else if constexpr (PRE) {
                    v[s] = spec_apply<OP>(v[s], hh[s], needY ? yy[needY ? s : 0] : z, needD ? dd1[needD ? s : 0] : z,
                                          needD ? dd2[needD ? s : 0] : z, mu, 2.0, acc);
                } else {
                    v[s] = spec_apply<OP>(v[s], p.H[hik], needY ? p.Y[hik] : z, needD ? p.D1[hik] : z,
                                          needD ? p.D2[hik] : z, mu, 2.0, acc);
                }
            }
        } else if constexpr (PRE) {
            const double2 z = make_double2(0.0, 0.0);
#pragma unroll
            for (int s = 0; s < 8; ++s)
                v[s] = spec_apply<OP>(v[s], hh[s], needY ? yy[needY ? s : 0] : z, needD ? dd1[needD ? s : 0] : z,
                                      needD ? dd2[needD ? s : 0] : z, mu, 2.0, acc);
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int l = t + s * T;
                const size_t hi = hbase + uidx(k, l);
                const double2 z = make_double2(0.0, 0.0);
                v[s] = spec_apply<OP>(v[s], p.H[hi], needY ? p.Y[hi] : z, needD ? p.D1[hi] : z,
                                      needD ? p.D2[hi] : z, mu, 2.0, acc);
            }
        }
        if constexpr (needY) {
            // deterministic block reduction of the accumulators
            constexpr int NT = RK * T;
            if constexpr (NT >= 64) {
                const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
                constexpr int NW = NT / 64;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    double a = acc[c];
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
                    if (lane == 0) red[c * 16 + w] = a;
                }
                __syncthreads();
                if (threadIdx.x < 3) {
                    double a = 0.0;
                    for (int ww = 0; ww < NW; ++ww) a += red[threadIdx.x * 16 + ww];
                    p.acc[((size_t)b * 3 + threadIdx.x) * gridDim.x + kb] = a;
                }
            } else {
                // partial wave: go through LDS (the exchange buffer is idle here)
                __syncthreads();
#pragma unroll
                for (int c = 0; c < 3; ++c) lds[c * NT + threadIdx.x] = acc[c];
                __syncthreads();
                if (threadIdx.x < 3) {
                    double a = 0.0;
                    for (int ww = 0; ww < NT; ++ww) a += lds[threadIdx.x * NT + ww];
                    p.acc[((size_t)b * 3 + threadIdx.x) * gridDim.x + kb] = a;
                }
            }
            __syncthreads();
        }
    }
    if (p.inv) fft_stages<LOG2N, 0, true>(v, t, p.tw, X);
    if (p.Sout) {
        double2 *__restrict__ out = p.Sout + ibase;
#pragma unroll
        for (int s = 0; s < 8; ++s) out[sidx(t + s * T)] = v[s];
    }
}

// ---------------------------------------------------------------------------
// unpack S -> U ((n1+1) x N, true spectrum rows 0..M/2)
// ---------------------------------------------------------------------------
__global__ void spec_unpack_kernel(const double2 *__restrict__ S, double2 *__restrict__ U, int n1, int N) {
    const int b = blockIdx.z;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;   // 0..n1
    const int l = blockIdx.y;
    if (k > n1) return;
    const double2 *s = S + (size_t)b * n1 * N;
    double2 *u = U + (size_t)b * (n1 + 1) * N;
    double2 r;
    if (k == 0 || k == n1) {
        const double2 P = s[(size_t)l * n1];
        const double2 Q = cconj(s[(size_t)((N - l) & (N - 1)) * n1]);
        if (k == 0) {
            r = cscale(cadd(P, Q), 0.5);
        } else {
            const double2 d = csub(P, Q);
            r = make_double2(0.5 * d.y, -0.5 * d.x);
        }
    } else {
        r = s[(size_t)l * n1 + k];
    }
    u[(size_t)l * (n1 + 1) + k] = r;
}

// direct DFT of the zero-padded taps (utils/resize.m:1-12): U[k,l] = sum_n c_n(k) wN^(l n) with
// c_n(k) = sum_m h[m,n] wM^(k m).  A thread owns one k: it forms its taille column sums c_n(k) once and then
// walks `lch` values of l with taille complex multiply-adds each (49 + 7 lch instead of 56 lch operations, in
// the summation order of the plain double loop).  The host picks lch = 1 for small spectra, where the number
// of threads in flight matters more than the operation count, and up to 16 for large ones.
constexpr int PSF_TMAX = 15;    // largest PSF size (sbtv.h)
// TT = taille known at compile time (7: every demo of the reference) -> fully unrolled, no predicates; TT = 0: any size
// up to three tap sets (the PSF and its parameter derivatives, SAPG with moving parameters) in ONE launch:
// blockIdx.z = set * batch + image
struct PsfSets {
    const double *taps[3];
    double2 *U[3];
};
template <int TT>
__global__ __launch_bounds__(64) void psf_spectrum_kernel(PsfSets ps, int taille, int batch, int n1, int M, int N,
                                                          const double2 *__restrict__ tw_M,
                                                          const double2 *__restrict__ tw_N, int lch, int tiled) {
    constexpr int TN = TT ? TT : PSF_TMAX;
    const int set = blockIdx.z / batch, b = blockIdx.z - set * batch;
    const double *__restrict__ taps = set == 0 ? ps.taps[0] : (set == 1 ? ps.taps[1] : ps.taps[2]);
    double2 *__restrict__ U = set == 0 ? ps.U[0] : (set == 1 ? ps.U[1] : ps.U[2]);
    // tiled layout U[(l/4)][k][l%4]: four lanes share one k and own the four slots of a 64-byte piece, so that a wave's
    // store is one contiguous kilobyte (one k per lane left every store instruction with 16 bytes in each of 64 lines:
    // 2.8 TB/s on the 134 MB the SAPG loop writes per iteration)
    const int k = (tiled == 1) ? blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2) : blockIdx.x * blockDim.x + threadIdx.x;
    if (k > n1) return;
    const double *h = taps + (size_t)b * taille * taille;
    double2 c[TN];
#pragma unroll
    for (int nn = 0; nn < TN; ++nn) {
        double2 cc = make_double2(0.0, 0.0);
        if (TT || nn < taille) {
#pragma unroll
            for (int m = 0; m < TN; ++m) {
                if (TT || m < taille) {
                    const double2 w = tw_M[(k * m) & (M - 1)];
                    const double hv = h[nn * taille + m];
                    cc.x += hv * w.x;
                    cc.y += hv * w.y;
                }
            }
        }
        c[nn] = cc;
    }
    double2 *ub = U + (size_t)b * (n1 + 1) * N;
    if (tiled == 1) {
        const int c4 = threadIdx.x & 3;
        const int t0 = blockIdx.y * lch, t1 = min(t0 + lch, N >> 2);
        for (int lt = t0; lt < t1; ++lt) {
            const int l = 4 * lt + c4;
            double2 acc = make_double2(0.0, 0.0);
#pragma unroll
            for (int nn = 0; nn < TN; ++nn)
                if (TT || nn < taille) acc = cadd(acc, cmul(c[nn], tw_N[(l * nn) & (N - 1)]));
            ub[((size_t)lt * (n1 + 1) + k) * 4 + c4] = acc;
        }
        return;
    }
    const int l0 = blockIdx.y * lch, l1 = min(l0 + lch, N);
    for (int l = l0; l < l1; ++l) {
        double2 acc = make_double2(0.0, 0.0);
#pragma unroll
        for (int nn = 0; nn < TN; ++nn)
            if (TT || nn < taille) acc = cadd(acc, cmul(c[nn], tw_N[(l * nn) & (N - 1)]));
        ub[tiled ? ((size_t)(l >> 2) * (n1 + 1) + k) * 4 + (l & 3) : (size_t)l * (n1 + 1) + k] = acc;
    }
}

#include "fft_wave.inc"
#ifdef SBTV_LAB
#include "fft_rows_sub.inc"
#endif
#include "fft_any.inc"

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
// Sizes that take the wave-granular kernels and the tiled spectrum layout: one wave per column needs M/2 = 64 x 8 or
// 64 x 16 points, the radix-16 row pass N = 1024 or 2048.  SBTV_FFT_WAVE=0 forces the workgroup kernels (A/B runs).
static inline bool wave_enabled() {
    static const bool on = [] {
        const char *e = getenv("SBTV_FFT_WAVE");
        return !(e && e[0] == '0');
    }();
    return on;
}
// row pass of the tiled sizes: the wave-granular kernel (SBTV_ROWS_KERNEL=wave) or the workgroup kernel (default: in
// the solver loops, where H / Y come from HBM, it is the faster one - profiles/r02_fft_insitu.md)
// (lab build only: it lost in the loop, profiles/r02_fft_lab.md; the default library does not carry the kernel)
static inline bool rows_wave() {
#ifdef SBTV_LAB
    static const bool on = [] {
        const char *e = getenv("SBTV_ROWS_KERNEL");
        return e && e[0] == 'w' && e[1] == 'a';
    }();
    return on;
#else
    return false;
#endif
}
// lab build, SBTV_ROWS_SUB=1: the row pass of the wave-granular sizes as four wave-local sub-transforms per row
// (rows_sub_kernel, fft_rows_sub.inc; operator spectra sub-row-major).  Round 3's fourth structural attempt on this pass: it
// ties with the software-pipelined kernel on every loop (profiles/r03_rows_sub_ab.md), so the default library keeps that one
static inline bool rows_sub() {
#ifdef SBTV_LAB
    static const bool on = [] {
        const char *e = getenv("SBTV_ROWS_SUB");
        return e && e[0] == '1';
    }();
    return on && !rows_wave();
#else
    return false;
#endif
}
// operator spectra tiled like S (default) or column-major U[l][k] (SBTV_U_TILED=0, the round-1 layout) in the wave mode
static inline bool u_tiled_wanted() {
#ifdef SBTV_LAB
    static const bool on = [] {
        const char *e = getenv("SBTV_U_TILED");
        return !(e && e[0] == '0');
    }();
    return on;
#else
    return true;
#endif
}
// software-pipelined row pass (rows_pipe_kernel) for the sizes of the wave mode; SBTV_ROWS_PIPE=0: the workgroup row
// kernel on the tiled layout instead (A/B runs: profiles/r02_rows_pipe.md)
static inline bool rows_pipe(int log2n) {
    (void)log2n;
#ifdef SBTV_LAB
    static const bool on = [] {
        const char *e = getenv("SBTV_ROWS_PIPE");
        return !(e && (e[0] == '0' || e[0] == '2'));
    }();
    return on;
#else
    return true;
#endif
}
// lab: SBTV_ROWS_PIPE=2 = the workgroup row kernel with TWO rows per 512-thread workgroup on the tiled layout (74 KB of LDS,
// no operand prefetch: two workgroups fit a CU) - round 3's "two independent workgroups per CU" experiment
static inline int rows_wg_rk() {
#ifdef SBTV_LAB
    static const int rk = [] {
        const char *e = getenv("SBTV_ROWS_PIPE");
        return (e && e[0] == '2') ? 2 : 4;
    }();
    return rk;
#else
    return 4;
#endif
}
#ifdef SBTV_LAB
static inline int rows_v(int dflt) {     // values per thread of the wave-granular row pass (tuning hook: SBTV_ROWS_V=8|16)
    static const int v = [] {
        const char *e = getenv("SBTV_ROWS_V");
        return e ? atoi(e) : 0;
    }();
    return (v == 8 || v == 16) ? v : dflt;
}
#endif

int fft_plan(sbtv_ctx *ctx, int M, int N, int batch, FftPlan *pl) {
    if (M < 2 || N < 2 || M > 4096 || N > 4096)
        return fail(ctx, SBTV_ERR_SIZE, "blur operator: 2 <= M, N <= 4096");
    pl->M = M;
    pl->N = N;
    pl->batch = batch;
    if (!is_pow2(M) || !is_pow2(N) || M < 16 || N < 16) {
        // arbitrary size (utils/resize.m pads the kernel to ANY image size): chirp-z path, full complex spectra
        pl->generic = 1;
        pl->wave = 0;
        pl->n1 = M;                       // so that 1 / (n1 N) is the inverse scale and S has n1 x N entries
        pl->u_ld = 0;
        pl->u_tiled = 0;
        pl->s_img = pl->u_img = (size_t)M * N;
        pl->tw_n1 = nullptr;
        SBTV_TRY(twiddle_get(ctx, M, &pl->tw_M));
        SBTV_TRY(twiddle_get(ctx, N, &pl->tw_N));
        return 0;
    }
    pl->generic = 0;
    pl->n1 = M / 2;
    pl->s_img = (size_t)pl->n1 * N;
    pl->wave = (wave_enabled() && (M == 1024 || M == 2048) && (N == 1024 || N == 2048)) ? 1 : 0;
    // operator spectra: row-major U[k][l] with a padded leading dimension for the wave-granular row kernel, the
    // column-major U[l][k] (leading dimension M/2 + 1, u_ld = 0) otherwise
    pl->u_ld = (pl->wave && rows_wave()) ? N + 16 : 0;
    pl->u_tiled = (pl->wave && rows_sub()) ? 2 : ((pl->wave && !rows_wave() && u_tiled_wanted()) ? 1 : 0);   // 2: U[k][l%4][l/4]
    pl->u_img = pl->u_ld ? (size_t)(pl->n1 + 1) * pl->u_ld : (size_t)(pl->n1 + 1) * N;
    SBTV_TRY(twiddle_get(ctx, pl->n1, &pl->tw_n1));
    SBTV_TRY(twiddle_get(ctx, M, &pl->tw_M));
    SBTV_TRY(twiddle_get(ctx, N, &pl->tw_N));
    return 0;
}

// columns per workgroup: as many as fit 256 threads, but small problems (fewer workgroups than CUs) are bound by
// the latency of one workgroup, not by throughput, so they get fewer columns per workgroup and more workgroups
static inline int cols_nseq(const FftPlan &pl) {
    const int T = pl.n1 / 8;
    int nseq = COLS_THREADS / T;
    if (nseq < 1) nseq = 1;
    if (nseq > pl.N) nseq = pl.N;
    // (never below one full wave per workgroup: the fused bookkeeping epilogue needs at least 6 threads)
    // The choice depends on the image size only, never on the batch: image k of a batch is then computed exactly
    // like image k alone (same partial-sum grouping), bit for bit.
    while (nseq > 1 && (nseq / 2) * T >= 64 && pl.N / nseq < 256) nseq >>= 1;
    return nseq;
}

template <int L>
static void launch_cols_fwd(sbtv_ctx *ctx, const FftPlan &pl, const double *x, const double *add, double2 *S,
                            const int *frozen, double *tvp) {
    const int nseq = cols_nseq(pl);
    hipLaunchKernelGGL(fft_cols_fwd_kernel<L>, dim3(pl.N / nseq, pl.batch), dim3(nseq * (pl.n1 / 8)), 0, ctx->stream,
                       x, add, S, pl.tw_n1, pl.tw_M, pl.N, frozen, tvp);
}
template <int L>
static void launch_cols_inv(sbtv_ctx *ctx, const FftPlan &pl, const double2 *S, double *x, double scale,
                            const int *frozen) {
    const int nseq = cols_nseq(pl);
    hipLaunchKernelGGL((fft_cols_inv_kernel<L, false>), dim3(pl.N / nseq, pl.batch), dim3(nseq * (pl.n1 / 8)), 0,
                       ctx->stream, S, x, pl.tw_n1, pl.tw_M, pl.N, scale, frozen, ColsPost{});
}
template <int L>
static void launch_cols_inv_post(sbtv_ctx *ctx, const FftPlan &pl, const double2 *S, double *x, double scale,
                                 const int *frozen, const ColsPost &post) {
    const int nseq = cols_nseq(pl);
    hipLaunchKernelGGL((fft_cols_inv_kernel<L, true>), dim3(pl.N / nseq, pl.batch), dim3(nseq * (pl.n1 / 8)), 0,
                       ctx->stream, S, x, pl.tw_n1, pl.tw_M, pl.N, scale, frozen, post);
}

#define SBTV_DISPATCH_LOG2(L, CALL)                                    \
    switch (L) {                                                       \
        case 3: CALL(3); break;                                        \
        case 4: CALL(4); break;                                        \
        case 5: CALL(5); break;                                        \
        case 6: CALL(6); break;                                        \
        case 7: CALL(7); break;                                        \
        case 8: CALL(8); break;                                        \
        case 9: CALL(9); break;                                        \
        case 10: CALL(10); break;                                      \
        case 11: CALL(11); break;                                      \
        default: return fail(ctx, SBTV_ERR_SIZE, "unsupported FFT length"); \
    }

// tvp != nullptr: also the periodic TV of x, as fft_cols_blocks(pl) partial sums per image (not with `add`, not on the
// arbitrary-size path: callers check fft_cols_tv_ok)
bool fft_cols_tv_ok(const FftPlan &pl) { return !pl.generic; }
int fft_cols_fwd_f(sbtv_ctx *ctx, const FftPlan &pl, const double *x, const double *add, double2 *S,
                   const int *frozen, double *tvp) {
    if (tvp && (pl.generic || add)) return fail(ctx, SBTV_ERR_BADARG, "fft_cols_fwd: TV partials not available for this call");
    if (pl.generic) return any_cols_fwd(ctx, pl, x, add, S, frozen);
    if (pl.wave) {
        const dim3 grid(pl.N / TW, pl.batch), block(64 * TW);
        if (pl.n1 == 1024)
            hipLaunchKernelGGL((cols_fwd_wave_kernel<10, 16>), grid, block, 0, ctx->stream, x, add, S, pl.tw_n1, pl.tw_M,
                               pl.N, frozen, tvp);
        else
            hipLaunchKernelGGL((cols_fwd_wave_kernel<9, 8>), grid, block, 0, ctx->stream, x, add, S, pl.tw_n1, pl.tw_M,
                               pl.N, frozen, tvp);
        SBTV_HIP(ctx, hipGetLastError());
        return 0;
    }
    const int L = ilog2(pl.n1);
#define CALL(LL) launch_cols_fwd<LL>(ctx, pl, x, add, S, frozen, tvp)
    SBTV_DISPATCH_LOG2(L, CALL)
#undef CALL
    SBTV_HIP(ctx, hipGetLastError());
    return 0;
}
int fft_cols_fwd(sbtv_ctx *ctx, const FftPlan &pl, const double *x, const double *add, double2 *S) {
    return fft_cols_fwd_f(ctx, pl, x, add, S, nullptr, nullptr);
}

int fft_cols_inv_f(sbtv_ctx *ctx, const FftPlan &pl, const double2 *S, double *x, double scale, const int *frozen) {
    if (pl.generic) return any_cols_inv(ctx, pl, S, x, scale, frozen, nullptr);
    if (pl.wave) {
        const dim3 grid(pl.N / TW, pl.batch), block(64 * TW);
        if (pl.n1 == 1024)
            hipLaunchKernelGGL((cols_inv_wave_kernel<10, 16, 0>), grid, block, 0, ctx->stream, S, x, pl.tw_n1,
                               pl.tw_M, pl.N, scale, frozen, ColsPost{});
        else
            hipLaunchKernelGGL((cols_inv_wave_kernel<9, 8, 0>), grid, block, 0, ctx->stream, S, x, pl.tw_n1, pl.tw_M,
                               pl.N, scale, frozen, ColsPost{});
        SBTV_HIP(ctx, hipGetLastError());
        return 0;
    }
    const int L = ilog2(pl.n1);
#define CALL(LL) launch_cols_inv<LL>(ctx, pl, S, x, scale, frozen)
    SBTV_DISPATCH_LOG2(L, CALL)
#undef CALL
    SBTV_HIP(ctx, hipGetLastError());
    return 0;
}
bool fft_cols_inv_step_ok(const FftPlan &pl) { return !pl.generic && pl.wave; }
int fft_cols_inv_step(sbtv_ctx *ctx, const FftPlan &pl, const double2 *S, double *y, double scale, double alpha,
                      const int *frozen) {
    if (!fft_cols_inv_step_ok(pl)) return fail(ctx, SBTV_ERR_BADARG, "fft_cols_inv_step: size not on the wave-granular path");
    ColsPost post;
    post.ystep = y;
    post.alpha = alpha;
    const dim3 grid(pl.N / TW, pl.batch), block(64 * TW);
    if (pl.n1 == 1024)
        hipLaunchKernelGGL((cols_inv_wave_kernel<10, 16, 8>), grid, block, 0, ctx->stream, S, (double *)nullptr, pl.tw_n1,
                           pl.tw_M, pl.N, scale, frozen, post);
    else
        hipLaunchKernelGGL((cols_inv_wave_kernel<9, 8, 8>), grid, block, 0, ctx->stream, S, (double *)nullptr, pl.tw_n1, pl.tw_M,
                           pl.N, scale, frozen, post);
    SBTV_HIP(ctx, hipGetLastError());
    return 0;
}
int fft_cols_inv_sub(sbtv_ctx *ctx, const FftPlan &pl, const double2 *S, double *x, double scale, const double *b, double *g) {
    if (!fft_cols_inv_step_ok(pl)) return fail(ctx, SBTV_ERR_BADARG, "fft_cols_inv_sub: size not on the wave-granular path");
    ColsPost post;
    post.bu_in = b;
    post.g = g;
    const dim3 grid(pl.N / TW, pl.batch), block(64 * TW);
    if (pl.n1 == 1024)
        hipLaunchKernelGGL((cols_inv_wave_kernel<10, 16, 64>), grid, block, 0, ctx->stream, S, x, pl.tw_n1, pl.tw_M, pl.N, scale,
                           (const int *)nullptr, post);
    else
        hipLaunchKernelGGL((cols_inv_wave_kernel<9, 8, 64>), grid, block, 0, ctx->stream, S, x, pl.tw_n1, pl.tw_M, pl.N, scale,
                           (const int *)nullptr, post);
    SBTV_HIP(ctx, hipGetLastError());
    return 0;
}
int fft_cols_inv_myula(sbtv_ctx *ctx, const FftPlan &pl, const double2 *S, double scale, double *X, const double *prox,
                       const double *Z, const double *sigma2_dev, double gam, double lamb, const RngArgs *rng,
                       const ProxArm *arm) {
    if (!fft_cols_inv_step_ok(pl)) return fail(ctx, SBTV_ERR_BADARG, "fft_cols_inv_myula: size not on the wave-granular path");
    if (!Z && !rng) return fail(ctx, SBTV_ERR_BADARG, "fft_cols_inv_myula: neither a noise array nor generator arguments");
    ColsPost post;
    post.ystep = X;
    post.mprox = prox;
    post.mZ = Z;
    post.msig2 = sigma2_dev;
    post.mgam = gam;
    post.mlamb = lamb;
    post.msq2g = sqrt(2 * gam);
    if (rng) post.mrng = *rng;
    if (arm) post.marm = *arm;
    const dim3 grid(pl.N / TW, pl.batch), block(64 * TW);
    if (pl.n1 == 1024)
        hipLaunchKernelGGL((cols_inv_wave_kernel<10, 16, 16>), grid, block, 0, ctx->stream, S, (double *)nullptr, pl.tw_n1,
                           pl.tw_M, pl.N, scale, (const int *)nullptr, post);
    else
        hipLaunchKernelGGL((cols_inv_wave_kernel<9, 8, 16>), grid, block, 0, ctx->stream, S, (double *)nullptr, pl.tw_n1, pl.tw_M,
                           pl.N, scale, (const int *)nullptr, post);
    SBTV_HIP(ctx, hipGetLastError());
    return 0;
}
int fft_cols_inv(sbtv_ctx *ctx, const FftPlan &pl, const double2 *S, double *x, double scale) {
    return fft_cols_inv_f(ctx, pl, S, x, scale, nullptr);
}
int fft_cols_blocks(const FftPlan &pl) { return (pl.wave || pl.generic) ? pl.N : pl.N / cols_nseq(pl); }
int fft_cols_inv_post(sbtv_ctx *ctx, const FftPlan &pl, const double2 *S, double *x, double scale, const int *frozen,
                      const ColsPost &post) {
    if ((!pl.wave && (post.bu_in || post.skip_x)) || (post.bu_in && !post.skip_x))
        return fail(ctx, SBTV_ERR_BADARG, "fft_cols_inv_post: bu_in + skip_x go together and need the wave-granular column pass");
    if (pl.generic) return any_cols_inv(ctx, pl, S, x, scale, frozen, &post);
    if (pl.wave) {
        const dim3 grid(pl.N / TW, pl.batch), block(64 * TW);
        if (post.skip_x && (!post.bu_in || post.bu_in == post.bu || post.xprev))
            return fail(ctx, SBTV_ERR_BADARG, "fft_cols_inv_post: skip_x needs a separate bu_in and no xprev");
        const int pm = 1 + (post.tru ? 2 : 0) + (post.xprev ? 4 : 0) + (post.skip_x ? 32 : 0);
#define SBTV_POST_CASE(PM_)                                                                                          \
    case PM_:                                                                                                        \
        if (pl.n1 == 1024)                                                                                           \
            hipLaunchKernelGGL((cols_inv_wave_kernel<10, 16, PM_>), grid, block, 0, ctx->stream, S, x, pl.tw_n1,       \
                               pl.tw_M, pl.N, scale, frozen, post);                                                  \
        else                                                                                                         \
            hipLaunchKernelGGL((cols_inv_wave_kernel<9, 8, PM_>), grid, block, 0, ctx->stream, S, x, pl.tw_n1,         \
                               pl.tw_M, pl.N, scale, frozen, post);                                                  \
        break;
        switch (pm) {
            SBTV_POST_CASE(1)
            SBTV_POST_CASE(3)
            SBTV_POST_CASE(5)
            SBTV_POST_CASE(7)
            SBTV_POST_CASE(33)
            SBTV_POST_CASE(35)
        }
#undef SBTV_POST_CASE
        SBTV_HIP(ctx, hipGetLastError());
        return 0;
    }
    const int L = ilog2(pl.n1);
#define CALL(LL) launch_cols_inv_post<LL>(ctx, pl, S, x, scale, frozen, post)
    SBTV_DISPATCH_LOG2(L, CALL)
#undef CALL
    SBTV_HIP(ctx, hipGetLastError());
    return 0;
}

// rows per workgroup of the row pass: 4 (64-byte segments) from N = 512 on, 8 below; N = 2048 may use
// 2 (SBTV_ROWS_RK=2: 512-thread workgroups, two per CU so load / FFT / store phases overlap)
static inline int rows_rk(const FftPlan &pl) {
#ifdef SBTV_LAB
    static const int rk2048 = [] {
        const char *e = getenv("SBTV_ROWS_RK");
        return (e && atoi(e) == 2) ? 2 : 4;
    }();
#else
    const int rk2048 = 4;
#endif
    const int N = pl.N;
    if (N == 4096) return 2;                 // two 4096-point rows fill the LDS exchange buffer (128 KB)
    if (N == 2048) return rk2048;
    const int rk = (N >= 512) ? 4 : 8;
    // small images: fewer rows per workgroup (more workgroups) when the grid of ONE image would leave most CUs
    // idle; a function of the image size only (see cols_nseq)
    const int small = (N == 512) ? 1 : (N == 256) ? 2 : rk;
    return (pl.n1 / rk < 128) ? small : rk;
}
bool fft_rows_csalsa_ok(const FftPlan &pl) {
    if (pl.generic) return false;
#ifdef SBTV_LAB
    if (pl.wave && rows_wave()) return false;   // the lab's wave-granular row kernel does not carry it
#endif
    return true;
}
int fft_rows_blocks(const FftPlan &pl) {
    if (pl.generic) return ANY_SPEC_BLOCKS;
    if (pl.wave) return (rows_wave() || rows_sub()) ? pl.n1 / 2 : (rows_pipe(0) ? pl.n1 / 4 : pl.n1 / rows_wg_rk());
    return pl.n1 / rows_rk(pl);
}

#ifdef SBTV_LAB
template <int L, int V>
static void launch_rows_wave(sbtv_ctx *ctx, const FftPlan &pl, const RowsParams &p) {
    const dim3 grid(pl.n1 / 2, pl.batch), block(2 * ((1 << L) / V));
    switch (p.op) {
        case OP_NONE: hipLaunchKernelGGL((rows_wave_kernel<L, V, OP_NONE>), grid, block, 0, ctx->stream, p); break;
        case OP_MUL_H: hipLaunchKernelGGL((rows_wave_kernel<L, V, OP_MUL_H>), grid, block, 0, ctx->stream, p); break;
        case OP_MUL_HC: hipLaunchKernelGGL((rows_wave_kernel<L, V, OP_MUL_HC>), grid, block, 0, ctx->stream, p); break;
        case OP_INVLS: hipLaunchKernelGGL((rows_wave_kernel<L, V, OP_INVLS>), grid, block, 0, ctx->stream, p); break;
        case OP_SALSA: hipLaunchKernelGGL((rows_wave_kernel<L, V, OP_SALSA>), grid, block, 0, ctx->stream, p); break;
        case OP_RESID: hipLaunchKernelGGL((rows_wave_kernel<L, V, OP_RESID>), grid, block, 0, ctx->stream, p); break;
        case OP_GRAD: hipLaunchKernelGGL((rows_wave_kernel<L, V, OP_GRAD>), grid, block, 0, ctx->stream, p); break;
        case OP_ATA: hipLaunchKernelGGL((rows_wave_kernel<L, V, OP_ATA>), grid, block, 0, ctx->stream, p); break;
        case OP_GRADF: hipLaunchKernelGGL((rows_wave_kernel<L, V, OP_GRADF>), grid, block, 0, ctx->stream, p); break;
        default: break;
    }
}
#endif

template <int L>
static void launch_rows_pipe(sbtv_ctx *ctx, const FftPlan &pl, const RowsParams &p0) {
    // chains sharing one spectrum set: fold the batch into grid.x (needs a row-block count per image that is a multiple of
    // the 8 XCDs, so that every XCD's share of the folded list holds whole groups of `batch` workgroups)
    static const bool fold_wanted = [] {
        const char *e = getenv("SBTV_ROWS_FOLD");
        return !(e && e[0] == '0');
    }();
    RowsParams p = p0;
    const bool fold = fold_wanted && p.shared_spec && pl.batch > 1 && ((pl.n1 / 4) % 8 == 0);
    p.fold = fold ? pl.batch : 0;
    const dim3 grid(fold ? (pl.n1 / 4) * pl.batch : pl.n1 / 4, fold ? 1 : pl.batch), block((1 << L) / 4);
#define SBTV_ROWS_OP(OP_)                                                                                     \
    case OP_: hipLaunchKernelGGL((rows_pipe_kernel<L, OP_>), grid, block, 0, ctx->stream, p); break;
    switch (p.op) {
        SBTV_ROWS_OP(OP_NONE)
        SBTV_ROWS_OP(OP_MUL_H)
        SBTV_ROWS_OP(OP_MUL_HC)
        SBTV_ROWS_OP(OP_INVLS)
        SBTV_ROWS_OP(OP_SALSA)
        SBTV_ROWS_OP(OP_RESID)
        SBTV_ROWS_OP(OP_GRAD)
        SBTV_ROWS_OP(OP_ATA)
        SBTV_ROWS_OP(OP_GRADF)
        SBTV_ROWS_OP(OP_CSALSA)
        default: break;
    }
#undef SBTV_ROWS_OP
}

#ifdef SBTV_LAB
template <int L>
static void launch_rows_sub(sbtv_ctx *ctx, const FftPlan &pl, const RowsParams &p) {
    const dim3 grid(pl.n1 / 2, pl.batch), block((1 << L) / 4);
#define SBTV_ROWS_OP(OP_)                                                                                     \
    case OP_: hipLaunchKernelGGL((rows_sub_kernel<L, OP_>), grid, block, 0, ctx->stream, p); break;
    switch (p.op) {
        SBTV_ROWS_OP(OP_NONE)
        SBTV_ROWS_OP(OP_MUL_H)
        SBTV_ROWS_OP(OP_MUL_HC)
        SBTV_ROWS_OP(OP_INVLS)
        SBTV_ROWS_OP(OP_SALSA)
        SBTV_ROWS_OP(OP_RESID)
        SBTV_ROWS_OP(OP_GRAD)
        SBTV_ROWS_OP(OP_ATA)
        SBTV_ROWS_OP(OP_GRADF)
        SBTV_ROWS_OP(OP_CSALSA)
        default: break;
    }
#undef SBTV_ROWS_OP
}
#endif

template <int L, int RK, bool TILED = false>
static void launch_rows(sbtv_ctx *ctx, const FftPlan &pl, const RowsParams &p) {
    const dim3 grid(pl.n1 / RK, pl.batch), block(RK * ((1 << L) / 8));
#define SBTV_ROWS_OP(OP_)                                                                                     \
    case OP_: hipLaunchKernelGGL((fft_rows_kernel<L, RK, OP_, TILED>), grid, block, 0, ctx->stream, p); break;
    switch (p.op) {
        SBTV_ROWS_OP(OP_NONE)
        SBTV_ROWS_OP(OP_MUL_H)
        SBTV_ROWS_OP(OP_MUL_HC)
        SBTV_ROWS_OP(OP_INVLS)
        SBTV_ROWS_OP(OP_SALSA)
        SBTV_ROWS_OP(OP_RESID)
        SBTV_ROWS_OP(OP_GRAD)
        SBTV_ROWS_OP(OP_ATA)
        SBTV_ROWS_OP(OP_GRADF)
        SBTV_ROWS_OP(OP_CSALSA)
        default: break;
    }
#undef SBTV_ROWS_OP
}

int fft_rows(sbtv_ctx *ctx, const FftPlan &pl, const double2 *Sin, double2 *Sout, const RowsArgs &a) {
    RowsParams p;
    p.Sin = Sin;
    p.Sout = Sout;
    p.tw = pl.tw_N;
    p.H = a.H;
    p.Y = a.Y;
    p.D1 = a.D1;
    p.D2 = a.D2;
    p.E = a.E;
    p.cs = a.cs;
    p.mu = a.mu;
    p.acc = a.acc;
    p.frozen = a.frozen;
    p.n1 = pl.n1;
    p.fwd = a.dir_fwd;
    p.inv = a.dir_inv;
    p.op = a.op;
    p.shared_spec = a.shared_spec;
    p.fold = 0;
    p.u_img = pl.u_img;
    p.u_ld = pl.u_ld;
    p.u_tiled = pl.u_tiled;
    if (a.op == OP_CSALSA && !fft_rows_csalsa_ok(pl)) return fail(ctx, SBTV_ERR_SIZE, "row pass: OP_CSALSA is not built for this plan");
    if (pl.generic) return any_rows(ctx, pl, p, Sout);
    const int L = ilog2(pl.N);
    if (L > 12) return fail(ctx, SBTV_ERR_SIZE, "row FFT: N must be <= 4096");
#ifdef SBTV_LAB
    if (pl.wave && rows_sub()) {
        if (L == 11) launch_rows_sub<11>(ctx, pl, p);
        else launch_rows_sub<10>(ctx, pl, p);
        SBTV_HIP(ctx, hipGetLastError());
        return 0;
    }
#endif
    if (pl.wave && !rows_wave()) {
        if (rows_pipe(L)) {
            // software-pipelined row pass (two row pairs per workgroup, loads in flight across the arithmetic)
            if (L == 11) launch_rows_pipe<11>(ctx, pl, p);
            else launch_rows_pipe<10>(ctx, pl, p);
            SBTV_HIP(ctx, hipGetLastError());
            return 0;
        }
#ifdef SBTV_LAB
        // the workgroup row kernel on the tiled layout: 4 rows x 4 columns = 256 contiguous bytes per access
        if (rows_wg_rk() == 2) {
            if (L == 11) launch_rows<11, 2, true>(ctx, pl, p);
            else launch_rows<10, 2, true>(ctx, pl, p);
        } else if (L == 11) launch_rows<11, 4, true>(ctx, pl, p);
        else launch_rows<10, 4, true>(ctx, pl, p);
        SBTV_HIP(ctx, hipGetLastError());
        return 0;
#endif
    }
#ifdef SBTV_LAB
    if (pl.wave) {
        // values per thread: 16 (radix-16 stages) at N = 2048, 8 at N = 1024 (measured: profiles/r02_fft_lab.md)
        const int V = rows_v(L == 11 ? 16 : 8);
        if (L == 11) {
            if (V == 8) launch_rows_wave<11, 8>(ctx, pl, p);
            else launch_rows_wave<11, 16>(ctx, pl, p);
        } else {
            if (V == 8) launch_rows_wave<10, 8>(ctx, pl, p);
            else launch_rows_wave<10, 16>(ctx, pl, p);
        }
        SBTV_HIP(ctx, hipGetLastError());
        return 0;
    }
#endif
    if (pl.N >= 512) {
        switch (L) {
            case 9:
                if (rows_rk(pl) == 1) launch_rows<9, 1>(ctx, pl, p);
                else launch_rows<9, 4>(ctx, pl, p);
                break;
            case 10: launch_rows<10, 4>(ctx, pl, p); break;
            case 11:
#ifdef SBTV_LAB
                if (rows_rk(pl) == 2) launch_rows<11, 2>(ctx, pl, p);
                else
#endif
                    launch_rows<11, 4>(ctx, pl, p);
                break;
            case 12: launch_rows<12, 2>(ctx, pl, p); break;
            default: break;
        }
    } else {
        switch (L) {
            case 4: launch_rows<4, 8>(ctx, pl, p); break;
            case 5: launch_rows<5, 8>(ctx, pl, p); break;
            case 6: launch_rows<6, 8>(ctx, pl, p); break;
            case 7: launch_rows<7, 8>(ctx, pl, p); break;
            case 8:
                if (rows_rk(pl) == 2) launch_rows<8, 2>(ctx, pl, p);
                else launch_rows<8, 8>(ctx, pl, p);
                break;
            default: return fail(ctx, SBTV_ERR_SIZE, "row FFT: N must be >= 16");
        }
    }
    SBTV_HIP(ctx, hipGetLastError());
    return 0;
}

int spec_unpack(sbtv_ctx *ctx, const FftPlan &pl, const double2 *S, double2 *U) {
    if (pl.generic) {          // the spectrum is already full: same layout as the operator spectra
        SBTV_HIP(ctx, hipMemcpyAsync(U, S, sizeof(double2) * pl.s_img * pl.batch, hipMemcpyDeviceToDevice, ctx->stream));
        return 0;
    }
    if (pl.wave) {
        hipLaunchKernelGGL(spec_unpack_tiled_kernel, dim3((pl.N + 63) / 64, pl.n1 + 1, pl.batch), dim3(64), 0,
                           ctx->stream, S, U, pl.n1, pl.N, pl.u_ld, pl.u_tiled);
        SBTV_HIP(ctx, hipGetLastError());
        return 0;
    }
    const int thr = 64;
    hipLaunchKernelGGL(spec_unpack_kernel, dim3((pl.n1 + 1 + thr - 1) / thr, pl.N, pl.batch), dim3(thr), 0,
                       ctx->stream, S, U, pl.n1, pl.N);
    SBTV_HIP(ctx, hipGetLastError());
    return 0;
}

// spectra of `nsets` (<= 3) tap sets of the same size: one launch on the default layouts, else one call per set
int psf_spectrum_sets(sbtv_ctx *ctx, const FftPlan &pl, const double *const *taps_dev, int taille, double2 *const *U, int nsets) {
    const int thr = 64;
    if (taille > PSF_TMAX) return fail(ctx, SBTV_ERR_PSF, "PSF larger than 15 x 15");
    if (nsets < 1 || nsets > 3) return fail(ctx, SBTV_ERR_BADARG, "psf_spectrum_sets: 1..3 tap sets");
    if (pl.generic) {
        for (int q = 0; q < nsets; ++q) SBTV_TRY(any_psf_spectrum(ctx, pl, taps_dev[q], taille, U[q]));
        return 0;
    }
    const size_t elems = (size_t)(pl.n1 + 1) * pl.N * pl.batch;
    int lch = (int)(elems >> 19);                       // 512^2, 1024^2: 1;  2048^2: 4;  8 x 1024^2: 8
    lch = lch < 1 ? 1 : (lch > 16 ? 16 : lch);
    if (pl.u_ld || pl.u_tiled == 2) {
        const dim3 gridw((pl.N + thr - 1) / thr, (pl.n1 + 1 + lch - 1) / lch, pl.batch);
        for (int q = 0; q < nsets; ++q) {
            if (taille == 7)
                hipLaunchKernelGGL(psf_spectrum_rowmajor_kernel<7>, gridw, dim3(thr), 0, ctx->stream, taps_dev[q], taille, U[q],
                                   pl.n1, pl.M, pl.N, pl.tw_M, pl.tw_N, lch, pl.u_ld ? pl.u_ld : pl.N, pl.u_tiled == 2);
            else
                hipLaunchKernelGGL(psf_spectrum_rowmajor_kernel<0>, gridw, dim3(thr), 0, ctx->stream, taps_dev[q], taille, U[q],
                                   pl.n1, pl.M, pl.N, pl.tw_M, pl.tw_N, lch, pl.u_ld ? pl.u_ld : pl.N, pl.u_tiled == 2);
        }
        SBTV_HIP(ctx, hipGetLastError());
        return 0;
    }
    PsfSets ps{};
    for (int q = 0; q < 3; ++q) {
        ps.taps[q] = taps_dev[q < nsets ? q : 0];
        ps.U[q] = U[q < nsets ? q : 0];
    }
    const dim3 grid = (pl.u_tiled == 1) ? dim3(((pl.n1 + 1) * 4 + thr - 1) / thr, ((pl.N >> 2) + lch - 1) / lch, pl.batch * nsets)
                                        : dim3((pl.n1 + 1 + thr - 1) / thr, (pl.N + lch - 1) / lch, pl.batch * nsets);
    if (taille == 7)
        hipLaunchKernelGGL(psf_spectrum_kernel<7>, grid, dim3(thr), 0, ctx->stream, ps, taille, pl.batch, pl.n1, pl.M, pl.N,
                           pl.tw_M, pl.tw_N, lch, pl.u_tiled);
    else
        hipLaunchKernelGGL(psf_spectrum_kernel<0>, grid, dim3(thr), 0, ctx->stream, ps, taille, pl.batch, pl.n1, pl.M, pl.N,
                           pl.tw_M, pl.tw_N, lch, pl.u_tiled);
    SBTV_HIP(ctx, hipGetLastError());
    return 0;
}
int psf_spectrum(sbtv_ctx *ctx, const FftPlan &pl, const double *taps_dev, int taille, double2 *U) {
    return psf_spectrum_sets(ctx, pl, &taps_dev, taille, &U, 1);
}

}  // namespace sbtv

using namespace sbtv;

extern "C" {

int sbtv_rfft2_packed(sbtv_ctx *ctx, const double *in, double *out, int M, int N, int batch, int inverse, int flags) {
    if (!ctx) return SBTV_ERR_BADARG;
    if (!in || !out || batch < 1) return fail(ctx, SBTV_ERR_BADARG, "rfft2_packed: bad arguments");
    SBTV_HIP(ctx, hipSetDevice(ctx->device));
    FftPlan pl;
    SBTV_TRY(fft_plan(ctx, M, N, batch, &pl));
    if (pl.generic)
        return fail(ctx, SBTV_ERR_SIZE, "rfft2_packed: the packed half-spectrum format exists for powers of two 16..4096 only");
    const size_t cnt = (size_t)M * N * batch;
    const double *ind = nullptr;
    double *outd = nullptr;
    SBTV_TRY(stage_in(ctx, "fft.in", in, cnt, flags, &ind));
    SBTV_TRY(stage_out_buf(ctx, "fft.out", out, cnt, flags, &outd));
    double2 *tmp = nullptr;
    SBTV_TRY(ws_get_t(ctx, "fft.tmp", cnt / 2, &tmp));
    RowsArgs a{};
    a.op = OP_NONE;
    // the documented packed format is the plain one ([l][k]); sizes on the wave-granular path keep the spectrum tiled
    // internally and convert at this (debug / test) boundary
    double2 *tmp2 = nullptr;
    if (pl.wave) SBTV_TRY(ws_get_t(ctx, "fft.tmp2", cnt / 2, &tmp2));
    auto relayout = [&](const double2 *src, double2 *dst, int to_tiled) -> int {
        hipLaunchKernelGGL(s_relayout_kernel, dim3(1024, batch), dim3(256), 0, ctx->stream, src, dst, pl.n1, N, to_tiled);
        SBTV_HIP(ctx, hipGetLastError());
        return 0;
    };
    if (!inverse) {
        SBTV_TRY(fft_cols_fwd(ctx, pl, ind, nullptr, tmp));
        a.dir_fwd = 1;
        a.dir_inv = 0;
        if (pl.wave) {
            SBTV_TRY(fft_rows(ctx, pl, tmp, tmp2, a));
            SBTV_TRY(relayout(tmp2, reinterpret_cast<double2 *>(outd), 0));
        } else {
            SBTV_TRY(fft_rows(ctx, pl, tmp, reinterpret_cast<double2 *>(outd), a));
        }
    } else {
        a.dir_fwd = 0;
        a.dir_inv = 1;
        if (pl.wave) {
            SBTV_TRY(relayout(reinterpret_cast<const double2 *>(ind), tmp2, 1));
            SBTV_TRY(fft_rows(ctx, pl, tmp2, tmp, a));
        } else {
            SBTV_TRY(fft_rows(ctx, pl, reinterpret_cast<const double2 *>(ind), tmp, a));
        }
        SBTV_TRY(fft_cols_inv(ctx, pl, tmp, outd, 1.0 / ((double)pl.n1 * N)));
    }
    SBTV_TRY(stage_out_copy(ctx, out, outd, cnt, flags));
    if (!(flags & SBTV_DEVICE_PTRS)) SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return canary_epilogue(ctx, 0);
}

int sbtv_A_wrapper(sbtv_ctx *ctx, const double *taps, int taille, const double *mu, const double *x, double *out,
                   int M, int N, int batch, int mode, int flags) {
    if (!ctx) return SBTV_ERR_BADARG;
    int op;
    switch (mode) {
        case 1: case 3: op = OP_MUL_H; break;
        case 2: op = OP_MUL_HC; break;
        case 9: op = OP_INVLS; break;
        default: return fail(ctx, SBTV_ERR_MODE, "The value of parameter mode must be 1 or 2 (or 3, 9).");
    }
    if (!taps || !x || !out || batch < 1) return fail(ctx, SBTV_ERR_BADARG, "A_wrapper: bad arguments");
    if (taille < 1 || taille > 15 || taille > M || taille > N)
        return fail(ctx, SBTV_ERR_PSF, "Mask does not fit inside array");
    if (mode == 9 && !mu) return fail(ctx, SBTV_ERR_MISSING_LS, "A_wrapper: mode 9 (invLS) needs mu");
    SBTV_HIP(ctx, hipSetDevice(ctx->device));
    FftPlan pl;
    SBTV_TRY(fft_plan(ctx, M, N, batch, &pl));
    const size_t cnt = (size_t)M * N * batch;
    const double *xd = nullptr;
    double *outd = nullptr;
    SBTV_TRY(stage_in(ctx, "op.in", x, cnt, flags, &xd));
    SBTV_TRY(stage_out_buf(ctx, "op.out", out, cnt, flags, &outd));
    double *taps_d = nullptr, *mu_d = nullptr;
    SBTV_TRY(ws_get_t(ctx, "op.taps", (size_t)batch * taille * taille, &taps_d));
    SBTV_HIP(ctx, hipMemcpyAsync(taps_d, taps, sizeof(double) * batch * taille * taille, hipMemcpyHostToDevice,
                                 ctx->stream));
    if (mu) {
        SBTV_TRY(ws_get_t(ctx, "op.mu", (size_t)batch, &mu_d));
        SBTV_HIP(ctx, hipMemcpyAsync(mu_d, mu, sizeof(double) * batch, hipMemcpyHostToDevice, ctx->stream));
    }
    double2 *Hs = nullptr, *S = nullptr;
    SBTV_TRY(ws_get_t(ctx, "op.H", (size_t)batch * pl.u_img, &Hs));
    SBTV_TRY(ws_get_t(ctx, "op.S", (size_t)batch * pl.s_img, &S));
    SBTV_TRY(psf_spectrum(ctx, pl, taps_d, taille, Hs));
    SBTV_TRY(fft_cols_fwd(ctx, pl, xd, nullptr, S));
    RowsArgs a{};
    a.dir_fwd = 1;
    a.dir_inv = 1;
    a.op = op;
    a.H = Hs;
    a.mu = mu_d;
    SBTV_TRY(fft_rows(ctx, pl, S, S, a));
    SBTV_TRY(fft_cols_inv(ctx, pl, S, outd, 1.0 / ((double)pl.n1 * N)));
    ctx->calls += batch;
    SBTV_TRY(stage_out_copy(ctx, out, outd, cnt, flags));
    if (!(flags & SBTV_DEVICE_PTRS)) SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return canary_epilogue(ctx, 0);
}

// ---- diagnostics: time ONE pass of the hot path on scratch data (bench.py's per-kernel roofline lines and the
// kernel tuning harness tools/fft_lab.py).  HIP events on the context stream, `reps` back-to-back launches after two
// untimed ones; *alg_bytes = algorithmic bytes of one launch.
namespace sbtv {
__global__ void diag_fill_kernel(double *__restrict__ x, size_t n, unsigned salt) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned h = ((unsigned)i * 2654435761u + salt) * 2246822519u;
        x[i] = (double)(h >> 8) * (255.0 / 16777216.0);          // [0, 255)
    }
}
static int diag_fill(sbtv_ctx *ctx, double *x, size_t n, unsigned salt) {
    hipLaunchKernelGGL(diag_fill_kernel, dim3(1024), dim3(256), 0, ctx->stream, x, n, salt);
    SBTV_HIP(ctx, hipGetLastError());
    return 0;
}
}  // namespace sbtv

int sbtv_diag_time_pass(sbtv_ctx *ctx, int pass, int M, int N, int batch, int reps, double *ms_avg, double *alg_bytes) {
    if (!ctx || !ms_avg || reps < 1 || batch < 1) return SBTV_ERR_BADARG;
    SBTV_HIP(ctx, hipSetDevice(ctx->device));
    FftPlan pl;
    SBTV_TRY(fft_plan(ctx, M, N, batch, &pl));
    const size_t P = (size_t)M * N, cnt = P * batch, spec = pl.u_img * batch;
    double *u = nullptr, *bu = nullptr, *g = nullptr, *tru = nullptr, *x = nullptr, *mu_d = nullptr, *taps_d = nullptr;
    double2 *S = nullptr, *H = nullptr, *Y = nullptr, *D1 = nullptr, *D2 = nullptr;
    SBTV_TRY(ws_get_t(ctx, "diag.u", cnt, &u));
    SBTV_TRY(ws_get_t(ctx, "diag.bu", cnt, &bu));
    SBTV_TRY(ws_get_t(ctx, "diag.g", cnt, &g));
    SBTV_TRY(ws_get_t(ctx, "diag.tru", cnt, &tru));
    SBTV_TRY(ws_get_t(ctx, "diag.x", cnt, &x));
    SBTV_TRY(ws_get_t(ctx, "diag.S", (size_t)batch * pl.s_img, &S));
    SBTV_TRY(ws_get_t(ctx, "diag.H", spec, &H));
    SBTV_TRY(ws_get_t(ctx, "diag.Y", spec, &Y));
    SBTV_TRY(ws_get_t(ctx, "diag.D1", spec, &D1));
    SBTV_TRY(ws_get_t(ctx, "diag.D2", spec, &D2));
    SBTV_TRY(ws_get_t(ctx, "diag.mu", (size_t)batch, &mu_d));
    SBTV_TRY(ws_get_t(ctx, "diag.taps", (size_t)batch * 49, &taps_d));
    const int nrb = fft_rows_blocks(pl), npb = fft_cols_blocks(pl);
    double *acc = nullptr, *postp = nullptr, *lam_d = nullptr;
    SBTV_TRY(ws_get_t(ctx, "diag.acc", (size_t)batch * 3 * nrb, &acc));
    SBTV_TRY(ws_get_t(ctx, "diag.post", (size_t)batch * 6 * npb, &postp));
    SBTV_TRY(ws_get_t(ctx, "diag.lam", (size_t)batch, &lam_d));
    {
        std::vector<double> h((size_t)batch * 49), m(batch, 0.003), l(batch, 6.0);
        const double p[3] = {0.4, 0.3, 0.0};
        SBTV_TRY(sbtv_psf_taps(SBTV_PSF_GAUSSIAN, 7, p, h.data(), nullptr, nullptr));
        for (int b = 1; b < batch; ++b) std::copy(h.begin(), h.begin() + 49, h.begin() + (size_t)b * 49);
        SBTV_HIP(ctx, hipMemcpyAsync(taps_d, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, ctx->stream));
        SBTV_HIP(ctx, hipMemcpyAsync(mu_d, m.data(), sizeof(double) * batch, hipMemcpyHostToDevice, ctx->stream));
        SBTV_HIP(ctx, hipMemcpyAsync(lam_d, l.data(), sizeof(double) * batch, hipMemcpyHostToDevice, ctx->stream));
        SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    SBTV_TRY(diag_fill(ctx, u, cnt, 1));
    SBTV_TRY(diag_fill(ctx, bu, cnt, 2));
    SBTV_TRY(diag_fill(ctx, g, cnt, 3));
    SBTV_TRY(diag_fill(ctx, tru, cnt, 4));
    SBTV_TRY(diag_fill(ctx, reinterpret_cast<double *>(S), cnt, 5));
    SBTV_TRY(psf_spectrum(ctx, pl, taps_d, 7, H));
    SBTV_TRY(diag_fill(ctx, reinterpret_cast<double *>(Y), 2 * spec, 6));
    SBTV_TRY(psf_spectrum(ctx, pl, taps_d, 7, D1));
    SBTV_TRY(psf_spectrum(ctx, pl, taps_d, 7, D2));
    ProxPlan pp;
    const bool pass_steps = (pass >= 11 && pass <= 15);      // one optimistic fused launch of pass - 10 Chambolle steps
    if (pass == 7 || pass == 8 || pass_steps) SBTV_TRY(prox_plan(ctx, M, N, batch, &pp, "diag.prox"));
    int launch_no = 0;
    const double inv_scale = 1.0 / ((double)pl.n1 * N);
    const double specb = 16.0 * (double)(pl.n1 + 1) * N * batch, img = 8.0 * (double)cnt;
    double bytes = 0.0;
    auto once = [&]() -> int {
        RowsArgs a{};
        a.H = H;
        a.Y = Y;
        a.D1 = D1;
        a.D2 = D2;
        a.mu = mu_d;
        a.acc = acc;
        switch (pass) {
            case 0: bytes = 3 * img; return fft_cols_fwd(ctx, pl, u, bu, S);
            case 1: a.dir_fwd = a.dir_inv = 1; a.op = OP_SALSA; bytes = 2 * img + 2 * specb; return fft_rows(ctx, pl, S, S, a);
            case 2: {
                ColsPost cp;
                cp.u = u; cp.bu = bu; cp.g = g; cp.tru = tru; cp.partials = postp;
                bytes = 7 * img;
                return fft_cols_inv_post(ctx, pl, S, x, inv_scale, nullptr, cp);
            }
            case 3: bytes = 2 * img; return fft_cols_inv(ctx, pl, S, x, inv_scale);
            case 4: a.dir_fwd = 1; a.op = OP_NONE; bytes = 2 * img; return fft_rows(ctx, pl, S, S, a);
            case 5: a.dir_fwd = a.dir_inv = 1; a.op = OP_GRAD; bytes = 2 * img + 4 * specb; return fft_rows(ctx, pl, S, S, a);
            case 6: a.dir_fwd = a.dir_inv = 1; a.op = OP_GRADF; bytes = 2 * img + 2 * specb; return fft_rows(ctx, pl, S, S, a);
            case 7:     // warm-started prox(10) with f, as in one SALSA outer iteration (2 fused launches + control)
                bytes = (40.0 * 10 + 8.0) * (double)cnt;
                SBTV_TRY(prox_reset(ctx, pp, lam_d, 1.0, 10, 1e-3, 0.249, true, nullptr));
                return prox_iterate(ctx, pp, g, 10, x, false);
            case 8:     // cold prox(25) with f, as in one SAPG / FISTA iteration
                bytes = (40.0 * 25 + 8.0) * (double)cnt;
                SBTV_TRY(prox_reset(ctx, pp, lam_d, 1.0, 25, 1e-3, 0.249, false, nullptr));
                return prox_iterate(ctx, pp, g, 25, x, true);
            default:
                if (pass_steps) {
                    // ONE fused launch of K warm-started steps without f and without control kernels (the first launch of
                    // a SALSA prox): t(K) = fixed (region load / store, fill and drain of the grid) + K x per-step
                    const int K = pass - 10;
                    if (!prox_spec_ok(pp, g, nullptr, K)) return fail(ctx, SBTV_ERR_BADARG, "sbtv_diag_time_pass: no fused launch for this shape");
                    bytes = 40.0 * (double)cnt;
                    return prox_iterate(ctx, pp, g, K, nullptr, false, 1, (launch_no++) & 1, nullptr);
                }
                return fail(ctx, SBTV_ERR_BADARG, "sbtv_diag_time_pass: unknown pass");
        }
    };
    if (pass == 7 || pass_steps) {
        // warm start: zero duals in slot 0 and a control block that SELECTS slot 0 (a fresh control buffer holds
        // garbage, and `keep_cur` below would then index the dual buffer with it)
        SBTV_TRY(prox_zero_duals(ctx, pp));
        SBTV_TRY(prox_reset(ctx, pp, lam_d, 1.0, 10, 1e-3, 0.249, false, nullptr));
    }
    for (int w = 0; w < 2; ++w) SBTV_TRY(once());
    SBTV_HIP(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    for (int r = 0; r < reps; ++r) SBTV_TRY(once());
    SBTV_HIP(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    SBTV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    SBTV_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]));
    *ms_avg = (double)ms / reps;
    if (alg_bytes) *alg_bytes = bytes;
    return canary_epilogue(ctx, 0);
}

}  // extern "C"
