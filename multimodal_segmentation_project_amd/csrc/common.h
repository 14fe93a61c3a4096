// common.h — shared device/host helpers for the mi3d HIP library (gfx950 / CDNA4 only).
//
// Internal activation layout: channels-last  [N][D][H][W][C]  ("NDHWC"), element type T in
// {float, bf16}, with an explicit channel stride `cs` (elements) so that a tensor can be a
// channel slice of a wider buffer (the skip-concat buffers: encoder output = channels [0,C),
// upconv output = channels [C,2C) of one [.., 2C] buffer; models/unet.py:84 becomes free).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

enum { MI3D_F32 = 0, MI3D_BF16 = 1 };

// ---- error plumbing (no C++ exception ever crosses the C ABI) ---------------------------------
void mi3d_set_error(const char* fmt, ...);
#define MI3D_CHECK_ARG(cond, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            mi3d_set_error(__VA_ARGS__);          \
            return -1;                            \
        }                                         \
    } while (0)
#define MI3D_HIP(call)                                                                   \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            mi3d_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (int)e_;                                                              \
        }                                                                                \
    } while (0)
#define MI3D_LAUNCH_CHECK() MI3D_HIP(hipGetLastError())
#define MI3D_TRY(expr)          \
    do {                        \
        int rc_ = (expr);       \
        if (rc_ != 0) return rc_; \
    } while (0)

// ---- route switches ------------------------------------------------------------------------------
// Every kernel-selection switch of the library lives in ONE struct.  It is filled from the environment exactly once, at
// the first use in the process (MI3D_<NAME>=<integer>; a variable that is set to something non-numeric counts as 1), so a
// captured hipGraph and later eager launches cannot disagree and no launch path calls getenv().  Tests and tools change a
// route through the ABI (mi3d_debug_set_route), not through setenv.  X(name, default)
#define MI3D_ROUTE_LIST(X)                                                                                          \
    X(force_direct, 0)      /* fp32-FMA kernels everywhere (no MFMA path) */                                        \
    X(no_planar, 0)         /* interleaved instead of planar skip/up halves at C = 16 */                            \
    X(no_persist, 0)        /* generic instead of persistent full-resolution conv kernels */                        \
    X(no_c1_mfma, 0)        /* first layer (Cin = 1) forward on the fp32 kernel */                                  \
    X(no_conv1_mfma, 0)     /* 1x1x1 head backward without the matrix cores */                                      \
    X(no_head_loss, 0)      /* head and loss as separate passes (logits / dlogits through memory) */                \
    X(no_pool_fuse, 0)      /* BatchNorm apply and MaxPool3d as two launches */                                     \
    X(no_small_bn, 0)       /* deep levels: statistics finished by a finalize launch, not the consumer's prologue */\
    X(no_defer_tail, 0)     /* split-K input gradients finished by their own pass */                                \
    X(no_pend_slabs, 0)     /* weight-gradient slab sums launched on their own */                                   \
    X(no_upbwd_carry, 0)    /* the decoder conv's slab sum is not carried across the transposed conv's backward */  \
    X(no_bwd_tail, 0)       /* split-K finish and slab sum of a fused backward as two launches */                   \
    X(no_fused_bwd, 0)      /* weight gradient and input gradient of a conv layer in two launches, every level */   \
    X(no_fused_bwd_p, 0)    /* ... full-resolution (persistent) layers only */                                      \
    X(no_fused_bwd_big, 0)  /* ... level-1 (16-wide tile) layers only */                                            \
    X(no_fused_upbwd, 0)    /* transposed conv: weight gradient and input gradient in two launches */               \
    X(api_unfused, 0)       /* mi3d_conv3_backward: stand-alone kernels instead of the step's fused launches */     \
    X(conv8, 1)             /* eight-wave forward conv for levels 1-4 (0: four-wave kernels) */                     \
    X(ks_target, 128)       /* split-K workgroup target, forward */                                                 \
    X(ks_target_bwd, 128)   /* split-K workgroup target, input gradient */                                          \
    X(fused_wg_target, 288) /* weight-gradient workgroups of a fused deep-level backward launch */                  \
    X(no_defer_wgrad, 0)    /* round 4: weight gradients stay on the data-gradient chain even with an aux stream */ \
    X(defer_mask, 7)        /* which weight gradients go to the aux stream: 1 decoder level 0, 2 decoder level 1, 4 deep levels */ \
    X(apply_on_load, 0)     /* MI3D_EXPERIMENTS builds only (round 4, measured slower): deep levels apply BatchNorm in the next conv's staging pass instead of a bn_apply / bn_bwd_apply launch */ \
    X(no_pool_splitk, 0)    /* 1: a split-K gradient of a pooled tensor is finished by its own pass, not inside the MaxPool3d backward */ \
    X(no_wide_store, 4)     /* bit mask of the kernels that keep 8-byte epilogue stores instead of 16-byte ones (v_permlane16_swap): 1 persistent conv, 2 eight-wave conv, 4 four-wave conv body of the fused backward, 8 transposed-conv forward; 15 = all.  In-process A/B of each site (us/step gained by the wide store): 0 / 5 / -3 / 7, so the four-wave body keeps its 8-byte stores */ \
    X(no_pool_pair, 0)      /* 1: MaxPool3d backward with one thread per window (rounds 1-3) instead of two */ \
    X(no_wgrad_xcd, 0)      /* 1: full-resolution weight gradients take tile = slab index (rounds 1-3) instead of XCD-contiguous tiles */ \
    X(no_upbwd_xcd_mix, 0)  /* 1: fused transposed-conv backward with the round-3 block mapping (even blocks weight gradient, odd data gradient: one kind per XCD) */ \
    X(opt_tail, 0)          /* 1: AdamW + weight re-pack of everything but the leading encoder blocks on the aux stream beside the end of the backward (TrainStep reads it; measured neutral: the aux stream is the long pole there) */ \
    X(wide_bn, 3)           /* round 4: the conv epilogue's BatchNorm partial rows are finished by the apply pass itself, no finalize launch: 1 = layers with <= 128 rows (level 2; every thin workgroup's prologue), 2 = also the layers with up to 1024 rows (levels 0-1) through wide_bn_wgs workgroups of 1024 threads; 0 = a finalize launch per layer (rounds 1-3) */ \
    X(wide_bn_wgs, 256)     /* workgroups of a wide BatchNorm pass */ \
    X(wide_min_rows, 129)   /* partial rows from which the wide kernel (instead of the thin workgroups' prologue) finishes the statistics */ \
    X(splitk_ticket, 1)     /* round 4: a split-K forward conv of a training step finishes itself (the last of a tile's ks workgroups sums the partials, stores y and the BatchNorm partial row); 0 = the bn_stats_splitk launch does (rounds 2-3) */ \
    X(conv_dma, 0)          /* MI3D_EXPERIMENTS builds only: LDS-DMA staging in the Cout = 16 persistent forward conv */
struct Mi3dRoutes {
#define MI3D_ROUTE_FIELD(name, dflt) int name = dflt;
    MI3D_ROUTE_LIST(MI3D_ROUTE_FIELD)
#undef MI3D_ROUTE_FIELD
};
const Mi3dRoutes& mi3d_routes();      // api.hip

#ifdef __HIPCC__
#define MI3D_HD __host__ __device__
#else
#define MI3D_HD
#endif
MI3D_HD static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- device helpers ---------------------------------------------------------------------------
#ifdef __HIPCC__
template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16>(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float v) { return (bf16)v; }

// round-trip through the storage type (what a consumer of the stored value will see)
template <typename T> __device__ __forceinline__ float round_to(float v) { return to_f<T>(from_f<T>(v)); }

// 8-wide vector load/store of T as floats (16 B for bf16, 2 x 16 B for f32).  p must be 16-B aligned.
template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&o)[8]);
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&o)[8]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}
template <> __device__ __forceinline__ void ld8<bf16>(const bf16* p, float (&o)[8]) {
    bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = (float)a[i];
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void st8<float>(float* p, const float (&v)[8]) {
    f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
}
template <> __device__ __forceinline__ void st8<bf16>(bf16* p, const float (&v)[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = (bf16)v[i];
    *reinterpret_cast<bf16x8*>(p) = a;
}

// VEC-generic (VEC = 8 vector path, VEC = 1 scalar path for odd channel counts)
template <typename T, int VEC> __device__ __forceinline__ void ldv(const T* p, float (&o)[VEC]) {
    if constexpr (VEC == 8) ld8<T>(p, o);
    else {
#pragma unroll
        for (int i = 0; i < VEC; i++) o[i] = to_f<T>(p[i]);
    }
}
template <typename T, int VEC> __device__ __forceinline__ void stv(T* p, const float (&v)[VEC]) {
    if constexpr (VEC == 8) st8<T>(p, v);
    else {
#pragma unroll
        for (int i = 0; i < VEC; i++) p[i] = from_f<T>(v[i]);
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
#endif  // __HIPCC__

// dtype dispatch for launchers: DISPATCH_T(dtype, T, { ... uses T ... })
#define DISPATCH_T(dtype, T, ...)                                  \
    do {                                                           \
        if ((dtype) == MI3D_F32) { typedef float T; __VA_ARGS__ }  \
        else { typedef bf16 T; __VA_ARGS__ }                       \
    } while (0)

// opt a kernel in to > 64 KB of dynamic LDS exactly once per process, safely from any host thread (the header promises
// "callable from any host thread": PyTorch runs backward on its autograd thread).  KPTR: parenthesise template kernels.
#ifdef __cplusplus
#include <mutex>
#define MI3D_SET_MAX_LDS_ONCE(KPTR, BYTES)                                                                   \
    do {                                                                                                     \
        static std::once_flag once_;                                                                         \
        hipError_t e_once_ = hipSuccess;                                                                     \
        std::call_once(once_, [&] {                                                                          \
            e_once_ = hipFuncSetAttribute(reinterpret_cast<const void*>(KPTR),                               \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES));         \
        });                                                                                                  \
        MI3D_HIP(e_once_);                                                                                   \
    } while (0)
#endif
