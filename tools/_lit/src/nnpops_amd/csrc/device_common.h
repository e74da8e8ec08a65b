// device_common.h -- helpers shared by the gfx950 kernels of libnnpops_hip.so.
//
// Written for CDNA4 only: 64-lane wavefronts are assumed everywhere (no warp-32 idioms, no
// portability macros).  The per-atom kernels run one atom per WAVE, 1-4 waves per workgroup, each
// wave with a private LDS slice: waves never talk to each other, so there are no block barriers,
// only wave_fence() below.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define NNPOPS_WAVE 64

namespace nnpops {

// ---------------------------------------------------------------------------------------------
// Periodic box, loaded once per wave from device memory (9 floats, rows = box vectors).
// Mirrors the reference's minimum-image rule: one round() per axis, applied z, y, x, using the
// diagonal element of each vector (reference src/ani/CpuANISymmetryFunctions.cpp:355-379).
// ---------------------------------------------------------------------------------------------
struct Box {
    float ax, bx, by, cx, cy, cz;     // lower-triangular entries actually used by the wrap
    float inv_x, inv_y, inv_z;
    bool triclinic;
};

__device__ __forceinline__ Box load_box(const float* __restrict__ box) {
    Box b;
    // uniform address -> scalar loads
    const float b00 = box[0], b01 = box[1], b02 = box[2];
    const float b10 = box[3], b11 = box[4], b12 = box[5];
    const float b20 = box[6], b21 = box[7], b22 = box[8];
    b.ax = b00; b.bx = b10; b.by = b11; b.cx = b20; b.cy = b21; b.cz = b22;
    // (v_rcp_f32, 1 ulp: the scaled coordinate only picks the image, see min_image)
    b.inv_x = __builtin_amdgcn_rcpf(b00); b.inv_y = __builtin_amdgcn_rcpf(b11); b.inv_z = __builtin_amdgcn_rcpf(b22);
    b.triclinic = (b01 != 0.f) | (b02 != 0.f) | (b10 != 0.f) | (b12 != 0.f) | (b20 != 0.f) | (b21 != 0.f);
    return b;
}

// The reference rounds half away from zero (round()); rintf (one v_rndne_f32 instead of seven instructions) differs
// from it only when a scaled component is exactly +-1/2, where both images are equally near and, the box being at
// least two cutoffs wide, at least a cutoff away: the pair is outside every list either way.
template <bool PERIODIC>
__device__ __forceinline__ void min_image(float& dx, float& dy, float& dz, const Box& b) {
    if (PERIODIC) {
        if (b.triclinic) {   // wave-uniform branch
            const float s3 = rintf(dz * b.inv_z);
            dx -= s3 * b.cx; dy -= s3 * b.cy; dz -= s3 * b.cz;
            const float s2 = rintf(dy * b.inv_y);
            dx -= s2 * b.bx; dy -= s2 * b.by;
            const float s1 = rintf(dx * b.inv_x);
            dx -= s1 * b.ax;
        } else {
            dx -= rintf(dx * b.inv_x) * b.ax;
            dy -= rintf(dy * b.inv_y) * b.by;
            dz -= rintf(dz * b.inv_z) * b.cz;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Wave-level primitives (64 lanes).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & (NNPOPS_WAVE - 1); }

// The wave-local fence between LDS producer and consumer phases of one wave (LDS operations of a wave execute
// in order; the fence stops the compiler from reordering them).  kWavesPerGroup is the LARGEST workgroup the
// per-atom kernels are launched with; the host picks 1, 2 or 4 waves per group so that LDS-limited occupancy is
// not lowered by the grouping (ani.hip: waves_per_group).
constexpr int kWavesPerGroup = 4;
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ int wave_in_group() { return threadIdx.x >> 6; }
__device__ __forceinline__ int wave_global_id() { return blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); }

// XCD-aware work order.  MI355X deals workgroups round-robin to its 8 XCDs (workgroup g runs on XCD g % 8), each
// with a private L2.  When neighbouring work items share data (atoms adjacent in cell order gather the same rows),
// give every XCD a CONTIGUOUS eighth of the ordered work instead of every eighth item: otherwise all eight L2s end
// up fetching everything.  Returns the position of this wave in the ordered sequence (a bijection on
// [0, gridDim.x * waves_per_group)).
__device__ __forceinline__ int xcd_contiguous_wave_id() {
    const int g = blockIdx.x, nwg = gridDim.x, xcd = g & 7;
    int start = 0;
    for (int c = 0; c < xcd; c++) start += (nwg - c + 7) >> 3;          // workgroups that land on the XCDs before mine
    return (start + (g >> 3)) * (blockDim.x >> 6) + (threadIdx.x >> 6);
}

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ int prefix_popc(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, NNPOPS_WAVE);
    return v;
}

// Sum over the wave, valid in LANE 63 ONLY: the same DPP ladder as wave_max_nonneg (6 vector instructions; wave_sum above
// costs six ds_bpermute with their address arithmetic and gives every lane the result, which a "lane 0 stores it" does
// not need).
__device__ __forceinline__ float wave_sum_lane63(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, false));     // row_shr:1
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, false));     // row_shr:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, false));     // row_shr:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, false));     // row_shr:8
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));     // row_bcast:15 -> rows 1, 3
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false));     // row_bcast:31 -> rows 2, 3
    return v;
}

// Largest value of a NON-NEGATIVE int over the wave, as a wave-uniform (scalar) value: a DPP scan inside each row
// of 16 lanes, two row broadcasts, the result read from lane 63 (no LDS traffic, 13 instructions).
__device__ __forceinline__ int wave_max_nonneg(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));     // row_shr:1
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));     // row_shr:2
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));     // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));     // row_shr:8
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));     // row_bcast:15 -> rows 1, 3
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));     // row_bcast:31 -> rows 2, 3
    return __builtin_amdgcn_readlane(v, 63);
}

// Write-through stores (sc1): the line stays valid in this XCD's L2 for the kernel that reads it next and is written to
// memory right away, so it is not part of the dirty-line write-back every kernel ends with.  A kernel that leaves tens
// of MB dirty (the neighbour build: 27 MB of rows and records) otherwise delays the start of the next one by ~5 us
// (profiles/r02f_timeline.txt); `nt` stores would also avoid that but evict the lines the next kernel wants.
typedef float f4_vec __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_wt(float4* p, const float4& v) {
    const f4_vec t = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ void store_wt(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store_wt(int* p, int v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

// Inclusive prefix sum of an int over the wave: the same DPP ladder (6 adds, no LDS round trips; __shfl_up costs a
// ds_bpermute and its address arithmetic per step).
__device__ __forceinline__ int wave_prefix_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2, 3
    return v;
}

// Fast single-instruction transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32 / v_sqrt_f32 /
// v_rsq_f32, ~1 ulp).  Used only in the per-triple / per-pair inner loops; per-neighbour
// quantities (cutoff function, distances) use the correctly-rounded library calls.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }

// sin(pi x) and cos(pi x) for x in [0, 1] -- the argument of the cosine cutoff, r / Rc.  With y = x - 1/2:
// cos(pi x) = -sin(pi y), sin(pi x) = cos(pi y), |y| <= 1/2, each a short polynomial in y^2 (least-squares fit
// on Chebyshev nodes, |error| <= 2e-7 absolute in fp32 Horner form -- the reference's own cosf(r * pi / Rc) is no
// closer to the exact value, its argument being rounded twice).  ~20 instructions for the pair; the library's
// sincospif is ~80, and the per-neighbour stages run it for every neighbour of every atom.
__device__ __forceinline__ void sincospi_unit(float x, float& s, float& c) {
    const float y = x - 0.5f, u = y * y;
    float ps = -0.00702838646247983f, pc = 0.0018400056287646294f;
    ps = fmaf(ps, u, 0.08205040544271469f);   pc = fmaf(pc, u, -0.025776328518986702f);
    ps = fmaf(ps, u, -0.5992522239685059f);   pc = fmaf(pc, u, 0.23532544076442719f);
    ps = fmaf(ps, u, 2.5501632690429688f);    pc = fmaf(pc, u, -1.3352622985839844f);
    ps = fmaf(ps, u, -5.167712688446045f);    pc = fmaf(pc, u, 4.058712005615234f);
    ps = fmaf(ps, u, 3.1415927410125732f);    pc = fmaf(pc, u, -4.934802055358887f);
    pc = fmaf(pc, u, 1.0f);
    c = -ps * y;                               // cos(pi x) = -sin(pi y)
    s = pc;                                    // sin(pi x) =  cos(pi y)
}

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kPi = 3.14159265358979323846f;

}  // namespace nnpops
