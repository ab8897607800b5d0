// Closest point on a triangulated surface — the search inside the ICP pre-alignment
// (reference: vtk_functions.py:12-29 -> vtkIterativeClosestPointTransform, whose inner loop asks a
// vtkCellLocator for the closest surface point of <= 1000 landmarks, 100 times).
//
// Exact search, no approximation: the answer is the minimum over ALL triangles of the exact
// point-triangle distance (ties: lowest triangle index), the same as a brute-force scan.  Pruning
// only removes triangles that provably cannot win:
//   build   triangles (polygons fan-triangulated) are sorted along a Morton curve of their centroids
//           (hipCUB radix sort) and cut into chunks of 64 consecutive ones, each with its bounding box;
//           coordinates are stored SoA so that a wave reads 64 consecutive triangles coalesced.
//           64 consecutive chunks form a super-chunk with its own box (two levels are enough: 500k
//           triangles = 7813 chunks = 123 super-chunks, two per lane).
//   query   ONE BLOCK (4 waves) PER QUERY POINT.  (1) the nearest super-chunk, then the nearest chunk inside it, by
//           point-box distance (one box per lane); that chunk is scanned first and yields an upper bound.
//           (2) one ballot over the super-chunk boxes, then per surviving super-chunk one ballot over its
//           64 chunk boxes; the surviving chunks are dealt round-robin to the 4 waves, which scan them one
//           triangle per lane, 4 chunks per step (loads issued together), re-deriving the bound after
//           every step.  Measured on a 250k pair: 41 chunk scans per landmark on average but 420 for the
//           farthest one, and the kernel lasts as long as its slowest query — hence 4 waves x 4 chunks
//           per step on that chain of dependent loads.
// The arithmetic of closest_on_triangle is Ericson's region walk, operation for operation the one in
// oracle/icp_port.py (compiled with -ffp-contract=off), so points and distances are bit-identical to it.
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <limits>

#include "pf_internal.h"

struct pf_surface {
    pf_ctx* ctx = nullptr;
    int64_t n_points = 0, n_faces = 0, n_tri = 0, n_chunks = 0;
    int32_t vpf = 0;
    double* tri = nullptr;       // SoA [9][n_tri]: ax ay az bx by bz cx cy cz, Morton order
    int32_t* tri_orig = nullptr; // [n_tri] sorted position -> triangle index (face * (vpf-2) + fan position)
    double* box = nullptr;       // [n_chunks][6] lo xyz, hi xyz
    double* sbox = nullptr;      // [n_super][6] boxes of 64 consecutive chunks
    int64_t n_super = 0;
};

namespace {

constexpr int PF_TRI_CHUNK = 64;  // one triangle per lane and scan

inline unsigned nblk(int64_t n) { return (unsigned)((n + PF_BLOCK - 1) / PF_BLOCK); }

struct Box3 {
    double lo[3], ext[3];
};

__device__ __forceinline__ unsigned spread10(unsigned v) {
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__device__ __forceinline__ void tri_vertices(const int32_t* __restrict__ faces, int32_t vpf, int64_t t, int32_t v[3]) {
    const int32_t per = vpf - 2;
    const int64_t f = t / per;
    const int32_t j = (int32_t)(t - f * per);
    v[0] = faces[f * vpf];
    v[1] = faces[f * vpf + j + 1];
    v[2] = faces[f * vpf + j + 2];
}

__global__ __launch_bounds__(PF_BLOCK) void k_tri_keys(const double* __restrict__ pts, const int32_t* __restrict__ faces,
                                                       int32_t vpf, int64_t n_tri, Box3 bb, unsigned* __restrict__ keys,
                                                       int32_t* __restrict__ vals) {
    const int64_t t = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (t >= n_tri) return;
    int32_t v[3];
    tri_vertices(faces, vpf, t, v);
    unsigned code = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double c = (pts[3 * (int64_t)v[0] + a] + pts[3 * (int64_t)v[1] + a] + pts[3 * (int64_t)v[2] + a]) / 3.0;
        double u = bb.ext[a] > 0.0 ? (c - bb.lo[a]) / bb.ext[a] : 0.0;
        u = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);  // also maps NaN to 0
        if (!(u == u)) u = 0.0;
        code |= spread10((unsigned)(u * 1023.0)) << a;
    }
    keys[t] = code;
    vals[t] = (int32_t)t;
}

__global__ __launch_bounds__(PF_BLOCK) void k_tri_gather(const double* __restrict__ pts, const int32_t* __restrict__ faces,
                                                         int32_t vpf, int64_t n_tri, const int32_t* __restrict__ order,
                                                         double* __restrict__ tri, int32_t* __restrict__ tri_orig) {
    const int64_t s = (int64_t)blockIdx.x * PF_BLOCK + threadIdx.x;
    if (s >= n_tri) return;
    const int32_t t = order[s];
    int32_t v[3];
    tri_vertices(faces, vpf, t, v);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int a = 0; a < 3; ++a) tri[(int64_t)(3 * c + a) * n_tri + s] = pts[3 * (int64_t)v[c] + a];
    tri_orig[s] = t;
}

// one wave per chunk
__global__ __launch_bounds__(PF_WAVE) void k_chunk_boxes(const double* __restrict__ tri, int64_t n_tri, double* __restrict__ box) {
    const int64_t c = blockIdx.x;
    const int lane = threadIdx.x;
    const double inf = std::numeric_limits<double>::infinity();
    double lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
    for (int64_t s = c * PF_TRI_CHUNK + lane; s < (c + 1) * PF_TRI_CHUNK && s < n_tri; s += PF_WAVE) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const double x = tri[(int64_t)k * n_tri + s];
            lo[k % 3] = fmin(lo[k % 3], x);  // fmin/fmax ignore NaN: a NaN vertex never widens a box
            hi[k % 3] = fmax(hi[k % 3], x);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int off = PF_WAVE / 2; off > 0; off >>= 1) {
            lo[a] = fmin(lo[a], __shfl_xor(lo[a], off, PF_WAVE));
            hi[a] = fmax(hi[a], __shfl_xor(hi[a], off, PF_WAVE));
        }
        if (lane == 0) {
            box[6 * c + a] = lo[a];
            box[6 * c + 3 + a] = hi[a];
        }
    }
}

// one wave per super-chunk: union of its 64 chunk boxes
__global__ __launch_bounds__(PF_WAVE) void k_super_boxes(const double* __restrict__ box, int64_t n_chunks, double* __restrict__ sbox) {
    const int64_t c = (int64_t)blockIdx.x * PF_WAVE + threadIdx.x;
    const double inf = std::numeric_limits<double>::infinity();
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double lo = c < n_chunks ? box[6 * c + a] : inf, hi = c < n_chunks ? box[6 * c + 3 + a] : -inf;
        for (int off = PF_WAVE / 2; off > 0; off >>= 1) {
            lo = fmin(lo, __shfl_xor(lo, off, PF_WAVE));
            hi = fmax(hi, __shfl_xor(hi, off, PF_WAVE));
        }
        if (threadIdx.x == 0) {
            sbox[6 * (int64_t)blockIdx.x + a] = lo;
            sbox[6 * (int64_t)blockIdx.x + 3 + a] = hi;
        }
    }
}

__device__ __forceinline__ double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// Ericson, Real-Time Collision Detection 5.1.5; same operation order as oracle/icp_port.py.
__device__ __forceinline__ void closest_on_triangle(const double p[3], const double a[3], const double b[3], const double c[3],
                                                    double out[3]) {
    double ab[3], ac[3], ap[3], bp[3], cp[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        ab[k] = b[k] - a[k];
        ac[k] = c[k] - a[k];
        ap[k] = p[k] - a[k];
        bp[k] = p[k] - b[k];
        cp[k] = p[k] - c[k];
    }
    const double d1 = dot3(ab, ap), d2 = dot3(ac, ap);
    if (d1 <= 0.0 && d2 <= 0.0) {
        out[0] = a[0], out[1] = a[1], out[2] = a[2];
        return;
    }
    const double d3 = dot3(ab, bp), d4 = dot3(ac, bp);
    if (d3 >= 0.0 && d4 <= d3) {
        out[0] = b[0], out[1] = b[1], out[2] = b[2];
        return;
    }
    const double vc = d1 * d4 - d3 * d2;
    if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) {
        const double v = d1 / (d1 - d3);
#pragma unroll
        for (int k = 0; k < 3; ++k) out[k] = a[k] + v * ab[k];
        return;
    }
    const double d5 = dot3(ab, cp), d6 = dot3(ac, cp);
    if (d6 >= 0.0 && d5 <= d6) {
        out[0] = c[0], out[1] = c[1], out[2] = c[2];
        return;
    }
    const double vb = d5 * d2 - d1 * d6;
    if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {
        const double w = d2 / (d2 - d6);
#pragma unroll
        for (int k = 0; k < 3; ++k) out[k] = a[k] + w * ac[k];
        return;
    }
    const double va = d3 * d6 - d5 * d4;
    if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {
        const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
#pragma unroll
        for (int k = 0; k < 3; ++k) out[k] = b[k] + w * (c[k] - b[k]);
        return;
    }
    const double denom = 1.0 / (va + vb + vc);
    const double v = vb * denom, w = vc * denom;
#pragma unroll
    for (int k = 0; k < 3; ++k) out[k] = (a[k] + ab[k] * v) + ac[k] * w;
}

__device__ __forceinline__ double box_dist2(const double p[3], const double* __restrict__ bx) {
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double below = bx[a] - p[a], above = p[a] - bx[3 + a];
        const double d = fmax(fmax(below, above), 0.0);
        s += d * d;
    }
    return s;
}

struct Best {
    double d2;
    int32_t orig;  // triangle index (tie-break: lowest)
    double pt[3];
};

__device__ __forceinline__ bool better(double d2, int32_t orig, double bd2, int32_t borig) {
    return d2 < bd2 || (d2 == bd2 && orig < borig);
}

// One triangle per lane from each of NB chunks: all loads are issued before the arithmetic, so the NB memory
// latencies overlap (the kernel is latency-bound: few waves, dependent loads).  Slots past the end of the mesh or
// repeated chunk indices re-evaluate a triangle that is already accounted for, which changes nothing.
template <int NB>
__device__ __forceinline__ void scan_chunks(const double* __restrict__ tri, const int32_t* __restrict__ tri_orig, int64_t n_tri,
                                            const int64_t (&chunk)[NB], int lane, const double p[3], Best& best) {
    double v[NB][9];
    int32_t orig[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        int64_t s = chunk[b] * PF_TRI_CHUNK + lane;
        s = s < n_tri ? s : n_tri - 1;
#pragma unroll
        for (int k = 0; k < 9; ++k) v[b][k] = tri[(int64_t)k * n_tri + s];
        orig[b] = tri_orig[s];
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const double a[3] = {v[b][0], v[b][1], v[b][2]}, bb[3] = {v[b][3], v[b][4], v[b][5]}, cc[3] = {v[b][6], v[b][7], v[b][8]};
        double q[3];
        closest_on_triangle(p, a, bb, cc, q);
        const double dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
        const double d2 = dx * dx + dy * dy + dz * dz;
        if (better(d2, orig[b], best.d2, best.orig)) {  // NaN distances compare false: never win
            best.d2 = d2;
            best.orig = orig[b];
            best.pt[0] = q[0], best.pt[1] = q[1], best.pt[2] = q[2];
        }
    }
}

__device__ __forceinline__ double wave_min(double v) {
    for (int off = PF_WAVE / 2; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, PF_WAVE));
    return v;
}

constexpr double PF_BOX_SLACK = 1.0 + 1e-9;  // the box test must never reject on a rounding error

// one block (PF_CLOSEST_WAVES waves) per query point
constexpr int PF_CLOSEST_WAVES = 4;  // waves per query point (8 measured slower: 0.29 vs 0.25 ms per 1000 landmarks)

__global__ __launch_bounds__(PF_CLOSEST_WAVES* PF_WAVE) void k_closest(const double* __restrict__ tri, const int32_t* __restrict__ tri_orig,
                                                      const double* __restrict__ box, const double* __restrict__ sbox,
                                                      int64_t n_tri, int64_t n_chunks, int64_t n_super,
                                                      const double* __restrict__ qry, int64_t n_qry, int32_t per_face,
                                                      double* __restrict__ out_pt, int32_t* __restrict__ out_face,
                                                      double* __restrict__ out_d2) {
    constexpr int NW = PF_CLOSEST_WAVES;
    constexpr int NB = 4;  // chunks scanned per step of a wave
    __shared__ double w_d2[NW], w_pt[NW][3];
    __shared__ int32_t w_orig[NW];
    const int lane = threadIdx.x & (PF_WAVE - 1), wave = threadIdx.x >> 6;
    const int64_t qi = blockIdx.x;
    const double p[3] = {qry[3 * qi], qry[3 * qi + 1], qry[3 * qi + 2]};
    const double inf = std::numeric_limits<double>::infinity();

    // (1) nearest super-chunk, nearest chunk inside it (by box distance; lowest index on ties) — every wave, same result
    auto wave_argmin = [&](double& d, int64_t& i) {
        for (int off = PF_WAVE / 2; off > 0; off >>= 1) {
            const double od = __shfl_xor(d, off, PF_WAVE);
            const int64_t oi = __shfl_xor(i, off, PF_WAVE);
            if (od < d || (od == d && oi < i)) d = od, i = oi;
        }
    };
    double nd = inf;
    int64_t ns = n_super;  // sentinel: none
    for (int64_t s = lane; s < n_super; s += PF_WAVE) {
        const double d = box_dist2(p, sbox + 6 * s);
        if (d < nd) nd = d, ns = s;
    }
    wave_argmin(nd, ns);
    int64_t c0 = n_chunks;
    if (ns < n_super) {
        c0 = ns * PF_WAVE + lane;
        nd = c0 < n_chunks ? box_dist2(p, box + 6 * c0) : inf;
        if (!(nd < inf)) c0 = n_chunks;
        wave_argmin(nd, c0);
    }
    Best best;
    best.d2 = inf, best.orig = 0x7fffffff, best.pt[0] = best.pt[1] = best.pt[2] = 0.0;
    if (c0 < n_chunks) {
        const int64_t one[1] = {c0};
        scan_chunks<1>(tri, tri_orig, n_tri, one, lane, p, best);
    }
    double bound = wave_min(best.d2);

    // (2) every chunk whose box is within the bound, super-chunk by super-chunk; of a super-chunk's surviving
    // chunks, wave w takes those at positions w, w+4, ... and scans NB of them per step
    const unsigned long long mine = (NW == 8 ? 0x0101010101010101ull : 0x1111111111111111ull) << wave;
    static_assert(NW == 4 || NW == 8, "chunk positions are dealt to 4 or 8 waves");
    for (int64_t sb = 0; sb < n_super; sb += PF_WAVE) {
        const int64_t s = sb + lane;
        unsigned long long smask = __ballot(s < n_super && box_dist2(p, sbox + 6 * s) <= bound * PF_BOX_SLACK);
        while (smask) {
            const int64_t ss = sb + __ffsll((long long)smask) - 1;
            smask &= smask - 1;
            if (box_dist2(p, sbox + 6 * ss) > bound * PF_BOX_SLACK) continue;  // the bound has shrunk since the ballot
            const int64_t c = ss * PF_WAVE + lane;
            unsigned long long mask = __ballot(c < n_chunks && c != c0 && box_dist2(p, box + 6 * c) <= bound * PF_BOX_SLACK) & mine;
            while (mask) {
                int64_t batch[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if (mask) {
                        batch[b] = ss * PF_WAVE + __ffsll((long long)mask) - 1;
                        mask &= mask - 1;
                    } else {
                        batch[b] = batch[0];
                    }
                }
                scan_chunks<NB>(tri, tri_orig, n_tri, batch, lane, p, best);
                bound = wave_min(best.d2);
                // drop the remaining chunks the tighter bound excludes (one box per lane, as in the ballot above)
                mask &= __ballot(c < n_chunks && box_dist2(p, box + 6 * c) <= bound * PF_BOX_SLACK);
            }
        }
    }

    // winner of the wave, then of the block: smallest distance, lowest triangle index
    double wd = best.d2;
    int32_t wo = best.orig;
    for (int off = PF_WAVE / 2; off > 0; off >>= 1) {
        const double od = __shfl_xor(wd, off, PF_WAVE);
        const int32_t oo = __shfl_xor(wo, off, PF_WAVE);
        if (better(od, oo, wd, wo)) wd = od, wo = oo;
    }
    if (lane == 0) w_d2[wave] = wd, w_orig[wave] = wo;
    if (best.orig == wo && best.d2 == wd && wo != 0x7fffffff) {  // the lane(s) holding (wd, wo) hold the same point
        w_pt[wave][0] = best.pt[0], w_pt[wave][1] = best.pt[1], w_pt[wave][2] = best.pt[2];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int win = 0;
        for (int w = 1; w < NW; ++w)
            if (better(w_d2[w], w_orig[w], w_d2[win], w_orig[win])) win = w;
        if (w_orig[win] != 0x7fffffff) {
            out_pt[3 * qi] = w_pt[win][0], out_pt[3 * qi + 1] = w_pt[win][1], out_pt[3 * qi + 2] = w_pt[win][2];
            out_face[qi] = w_orig[win] / per_face;
            out_d2[qi] = w_d2[win];
        } else {  // NaN query or no finite triangle
            out_pt[3 * qi] = out_pt[3 * qi + 1] = out_pt[3 * qi + 2] = __longlong_as_double(0x7ff8000000000000ll);
            out_face[qi] = -1;
            out_d2[qi] = inf;
        }
    }
}

}  // namespace

extern "C" {

void pf_surface_free(pf_surface* s) {
    if (!s) return;
    hipSetDevice(s->ctx->device);
    hipStreamSynchronize(s->ctx->stream);
    hipStream_t st = s->ctx->stream;
    pf_free(st, s->tri);
    pf_free(st, s->tri_orig);
    pf_free(st, s->box);
    pf_free(st, s->sbox);
    delete s;
}

int pf_surface_create(pf_ctx* ctx, const double* pts, int64_t n, const int32_t* faces, int64_t n_faces, int32_t vpf,
                      pf_surface** out) {
    PF_CHECK(ctx && pts && faces && out, PF_E_ARG, "pf_surface_create: NULL argument");
    PF_CHECK(n > 0 && n < ((int64_t)1 << 31), PF_E_ARG, "pf_surface_create: n = %lld out of range", (long long)n);
    PF_CHECK(vpf >= 3 && vpf <= 16 && n_faces > 0 && n_faces * (vpf - 2) < ((int64_t)1 << 31), PF_E_ARG,
             "pf_surface_create: faces %lld x %d out of range", (long long)n_faces, vpf);
    for (int64_t i = 0; i < n_faces * vpf; ++i)
        PF_CHECK(faces[i] >= 0 && faces[i] < n, PF_E_ARG, "pf_surface_create: face %lld references vertex %d of %lld",
                 (long long)(i / vpf), faces[i], (long long)n);
    Box3 bb;
    for (int a = 0; a < 3; ++a) {
        double lo = std::numeric_limits<double>::infinity(), hi = -lo;
        for (int64_t i = 0; i < n; ++i) {
            const double x = pts[3 * i + a];
            if (x < lo) lo = x;
            if (x > hi) hi = x;
        }
        bb.lo[a] = lo;
        bb.ext[a] = hi - lo;
        if (!(bb.ext[a] > 0.0) || !std::isfinite(bb.ext[a])) bb.ext[a] = 0.0;
    }
    PF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    pf_surface* s = new pf_surface();
    s->ctx = ctx;
    s->n_points = n, s->n_faces = n_faces, s->vpf = vpf;
    s->n_tri = n_faces * (vpf - 2);
    s->n_chunks = (s->n_tri + PF_TRI_CHUNK - 1) / PF_TRI_CHUNK;
    s->n_super = (s->n_chunks + PF_WAVE - 1) / PF_WAVE;
    const int64_t T = s->n_tri;
    double* d_pts = nullptr;
    int32_t* d_faces = nullptr;
    unsigned *k0 = nullptr, *k1 = nullptr;
    int32_t *v0 = nullptr, *v1 = nullptr;
    void* tmp = nullptr;
    size_t need = 0;
    hipError_t e = hipSuccess;
    do {
        if ((e = pf_malloc(st, (void**)&d_pts, sizeof(double) * 3 * n)) != hipSuccess) break;
        if ((e = pf_malloc(st, (void**)&d_faces, sizeof(int32_t) * n_faces * vpf)) != hipSuccess) break;
        if ((e = pf_malloc(st, (void**)&k0, sizeof(unsigned) * T)) != hipSuccess) break;
        if ((e = pf_malloc(st, (void**)&k1, sizeof(unsigned) * T)) != hipSuccess) break;
        if ((e = pf_malloc(st, (void**)&v0, sizeof(int32_t) * T)) != hipSuccess) break;
        if ((e = pf_malloc(st, (void**)&v1, sizeof(int32_t) * T)) != hipSuccess) break;
        if ((e = pf_malloc(st, (void**)&s->tri, sizeof(double) * 9 * T)) != hipSuccess) break;
        if ((e = pf_malloc(st, (void**)&s->tri_orig, sizeof(int32_t) * T)) != hipSuccess) break;
        if ((e = pf_malloc(st, (void**)&s->box, sizeof(double) * 6 * s->n_chunks)) != hipSuccess) break;
        if ((e = pf_malloc(st, (void**)&s->sbox, sizeof(double) * 6 * s->n_super)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d_pts, pts, sizeof(double) * 3 * n, hipMemcpyHostToDevice, st)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d_faces, faces, sizeof(int32_t) * n_faces * vpf, hipMemcpyHostToDevice, st)) != hipSuccess) break;
        k_tri_keys<<<nblk(T), PF_BLOCK, 0, st>>>(d_pts, d_faces, vpf, T, bb, k0, v0);
        if ((e = hipGetLastError()) != hipSuccess) break;
        if ((e = hipcub::DeviceRadixSort::SortPairs(nullptr, need, k0, k1, v0, v1, (int)T, 0, 30, st)) != hipSuccess) break;
        if ((e = pf_malloc(st, &tmp, need)) != hipSuccess) break;
        if ((e = hipcub::DeviceRadixSort::SortPairs(tmp, need, k0, k1, v0, v1, (int)T, 0, 30, st)) != hipSuccess) break;
        k_tri_gather<<<nblk(T), PF_BLOCK, 0, st>>>(d_pts, d_faces, vpf, T, v1, s->tri, s->tri_orig);
        k_chunk_boxes<<<(unsigned)s->n_chunks, PF_WAVE, 0, st>>>(s->tri, T, s->box);
        k_super_boxes<<<(unsigned)s->n_super, PF_WAVE, 0, st>>>(s->box, s->n_chunks, s->sbox);
        if ((e = hipGetLastError()) != hipSuccess) break;
        e = hipStreamSynchronize(st);  // the host arrays may go away after the call
    } while (0);
    pf_free(st, d_pts);
    pf_free(st, d_faces);
    pf_free(st, k0);
    pf_free(st, k1);
    pf_free(st, v0);
    pf_free(st, v1);
    pf_free(st, tmp);
    if (e != hipSuccess) {
        pf_set_error("pf_surface_create: %s", hipGetErrorString(e));
        pf_surface_free(s);
        return PF_E_HIP;
    }
    *out = s;
    return PF_OK;
}

int pf_surface_closest(pf_surface* s, const double* qry, int64_t n_qry, double* out_pts, int32_t* out_face, double* out_d2) {
    PF_CHECK(s && (qry || n_qry == 0), PF_E_ARG, "pf_surface_closest: NULL argument");
    PF_CHECK(n_qry >= 0 && n_qry < ((int64_t)1 << 31), PF_E_ARG, "pf_surface_closest: n_qry = %lld out of range", (long long)n_qry);
    if (n_qry == 0) return PF_OK;
    PF_HIP(hipSetDevice(s->ctx->device));
    hipStream_t st = s->ctx->stream;
    double *d_q = nullptr, *d_pt = nullptr, *d_d2 = nullptr;
    int32_t* d_face = nullptr;
    hipError_t e = hipSuccess;
    do {
        if ((e = pf_malloc(st, (void**)&d_q, sizeof(double) * 3 * n_qry)) != hipSuccess) break;
        if ((e = pf_malloc(st, (void**)&d_pt, sizeof(double) * 3 * n_qry)) != hipSuccess) break;
        if ((e = pf_malloc(st, (void**)&d_d2, sizeof(double) * n_qry)) != hipSuccess) break;
        if ((e = pf_malloc(st, (void**)&d_face, sizeof(int32_t) * n_qry)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d_q, qry, sizeof(double) * 3 * n_qry, hipMemcpyHostToDevice, st)) != hipSuccess) break;
        k_closest<<<(unsigned)n_qry, PF_CLOSEST_WAVES * PF_WAVE, 0, st>>>(s->tri, s->tri_orig, s->box, s->sbox, s->n_tri, s->n_chunks, s->n_super, d_q, n_qry,
                                               s->vpf - 2, d_pt, d_face, d_d2);
        if ((e = hipGetLastError()) != hipSuccess) break;
        if (out_pts && (e = hipMemcpyAsync(out_pts, d_pt, sizeof(double) * 3 * n_qry, hipMemcpyDeviceToHost, st)) != hipSuccess) break;
        if (out_face && (e = hipMemcpyAsync(out_face, d_face, sizeof(int32_t) * n_qry, hipMemcpyDeviceToHost, st)) != hipSuccess) break;
        if (out_d2 && (e = hipMemcpyAsync(out_d2, d_d2, sizeof(double) * n_qry, hipMemcpyDeviceToHost, st)) != hipSuccess) break;
        e = hipStreamSynchronize(st);
    } while (0);
    pf_free(st, d_q);
    pf_free(st, d_pt);
    pf_free(st, d_d2);
    pf_free(st, d_face);
    if (e != hipSuccess) {
        pf_set_error("pf_surface_closest: %s", hipGetErrorString(e));
        return PF_E_HIP;
    }
    return PF_OK;
}

}  // extern "C"
