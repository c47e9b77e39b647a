// ehr_pose_core.h -- device code shared by the stand-alone pose kernels (ehr_pose.hip) and the merged step kernels
// (ehr_fused.hip): se3 exponential with forward-mode partials, projection, 4x4 products, the pose backward and Adam.
// Reference formulas: /root/reference/easyhec/utils/pytorch3d_se3.py:12-41, :46-130, :218-245;
// /root/reference/easyhec/utils/nvdiffrast_utils.py:5-11; /root/reference/easyhec/solver/build.py:12-29.
#pragma once
#include "ehr_device.h"

namespace ehr {

// ---- forward-mode scalar with N partials ----------------------------------------------------------------------
template <int N>
struct Dual {
    float v;
    float d[N];
};
template <int N>
__device__ __forceinline__ Dual<N> dconst(float c) {
    Dual<N> r;
    r.v = c;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = 0.f;
    return r;
}
template <int N>
__device__ __forceinline__ Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r;
    r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = a.d[i] + b.d[i];
    return r;
}
template <int N>
__device__ __forceinline__ Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r;
    r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = a.d[i] - b.d[i];
    return r;
}
template <int N>
__device__ __forceinline__ Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r;
    r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
template <int N>
__device__ __forceinline__ Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r;
    float inv = 1.f / b.v;
    r.v = a.v * inv;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
template <int N>
__device__ __forceinline__ Dual<N> dneg(const Dual<N>& a) {
    return dconst<N>(0.f) - a;
}
template <int N>
__device__ __forceinline__ Dual<N> dsqrt(const Dual<N>& a) {
    Dual<N> r;
    r.v = sqrtf(a.v);
    float k = 0.5f / r.v;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = a.d[i] * k;
    return r;
}
template <int N>
__device__ __forceinline__ Dual<N> dsin(const Dual<N>& a) {
    Dual<N> r;
    r.v = sinf(a.v);
    float c = cosf(a.v);
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = a.d[i] * c;
    return r;
}
template <int N>
__device__ __forceinline__ Dual<N> dcos(const Dual<N>& a) {
    Dual<N> r;
    r.v = cosf(a.v);
    float s = -sinf(a.v);
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = a.d[i] * s;
    return r;
}
// torch.clamp(x, min=eps): value max(x, eps), gradient passes only where x >= eps
template <int N>
__device__ __forceinline__ Dual<N> dclamp_min(const Dual<N>& a, float eps) {
    return (a.v >= eps) ? a : dconst<N>(eps);
}

typedef Dual<6> D6;

// Tc (row-major 4x4, the usual [[R, t], [0, 1]]) and partials from dof = [log_translation, log_rotation].
// N = 6: all six partials (slot i = d/d dof_i).  N = 1: only the partial w.r.t. dof[which].
template <int N>
__device__ void se3_exp_dual(const float* dof, float eps, Dual<N> T[16], int which = 0) {
    Dual<N> x[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        x[i] = dconst<N>(dof[i]);
        if (N == 6) x[i].d[i % N] = 1.f;
        else if (i == which) x[i].d[0] = 1.f;
    }
    const Dual<N> u0 = x[0], u1 = x[1], u2 = x[2], w0 = x[3], w1 = x[4], w2 = x[5];
    Dual<N> nrms = w0 * w0 + w1 * w1 + w2 * w2;
    Dual<N> th = dsqrt(dclamp_min(nrms, eps));
    Dual<N> one = dconst<N>(1.f), zero = dconst<N>(0.f);
    Dual<N> inv = one / th;
    Dual<N> fac1 = inv * dsin(th);
    Dual<N> fac2 = inv * inv * (one - dcos(th));
    // hat(w) and its square
    Dual<N> K[9] = {zero, dneg(w2), w1, w2, zero, dneg(w0), dneg(w1), w0, zero};
    Dual<N> K2[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) K2[3 * r + c] = K[3 * r] * K[c] + K[3 * r + 1] * K[3 + c] + K[3 * r + 2] * K[6 + c];
    Dual<N> fv1 = (one - dcos(th)) / (th * th);
    Dual<N> fv2 = (th - dsin(th)) / (th * th * th);
    Dual<N> u[3] = {u0, u1, u2};
    for (int r = 0; r < 3; r++) {
        Dual<N> t = zero;
        for (int c = 0; c < 3; c++) {
            Dual<N> I = dconst<N>(r == c ? 1.f : 0.f);
            T[4 * r + c] = fac1 * K[3 * r + c] + fac2 * K2[3 * r + c] + I;
            Dual<N> V = I + K[3 * r + c] * fv1 + K2[3 * r + c] * fv2;
            t = t + V * u[c];
        }
        T[4 * r + 3] = t;
    }
    T[12] = zero;
    T[13] = zero;
    T[14] = zero;
    T[15] = one;
}

__device__ __forceinline__ void mat4_mul(const float* A, const float* B, float* C) {
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            float s = A[4 * r] * B[c];
            s = fmaf(A[4 * r + 1], B[4 + c], s);
            s = fmaf(A[4 * r + 2], B[8 + c], s);
            s = fmaf(A[4 * r + 3], B[12 + c], s);
            C[4 * r + c] = s;
        }
}

__device__ __forceinline__ void projection(const float* K, int H, int W, float n, float f, float* P) {
    float fu = K[0], fv = K[4], cu = K[2], cv = K[5];
    for (int i = 0; i < 16; i++) P[i] = 0.f;
    P[0] = 2.f * fu / (float)W;
    P[2] = -2.f * cu / (float)W + 1.f;
    P[5] = 2.f * fv / (float)H;
    P[6] = 2.f * cv / (float)H - 1.f;
    P[10] = -(f + n) / (f - n);
    P[11] = -2.f * f * n / (f - n);
    P[14] = -1.f;
}


// MVP[b,l] = proj @ (opencv2blender @ (Tc @ link_pose))   (rb_solver.py:63; nvdiffrast_renderer.py:35,37)
__device__ __forceinline__ void mvp_from_pose(const float* Tc, const float* P, const float* lp, float* C) {
    float A[16];
    mat4_mul(Tc, lp, A);
    for (int c = 0; c < 4; c++) {  // opencv2blender = diag(1,-1,-1,1)
        A[4 + c] = -A[4 + c];
        A[8 + c] = -A[8 + c];
    }
    mat4_mul(P, A, C);
}

// d(sum_b loss_b)/d dof, the loss sum and the frame count from d loss_b / d MVP[b,l]; one 256-thread workgroup,
// fixed-order reductions (a shuffle butterfly per wave, then the four waves in order).  S: LDS double [4][17].
// get_g(i, G) fills the 16 floats of d loss / d MVP for (view, link) pair i, get_loss(b) returns frame b's loss: the
// standalone kernel reads them from global memory, the in-kernel finish of the fused op straight from its accumulators
// (no store -> barrier -> reload round trip).  red_lds (optional, LDS float[8]) receives a copy of red for a
// following pose_adam_apply in the same workgroup.
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += wave_xor<32>(v);
    v += wave_xor<16>(v);
    v += wave_xor<8>(v);
    v += wave_xor<4>(v);
    v += wave_xor<2>(v);
    v += wave_xor<1>(v);
    return v;
}
template <class GetG, class GetLoss>
__device__ __forceinline__ void pose_backward_block_t(GetG get_g, GetLoss get_loss, const float* __restrict__ K,
                                                      const float* __restrict__ link_poses,
                                                      const float* __restrict__ tc_jac, int B, int L, int H, int W,
                                                      float n, float f, float* __restrict__ red, double (*S)[17],
                                                      float* red_lds, const double* la_pre,
                                                      bool la_lanes_0_32, float (*Js_lds)[16]) {
    // (la_pre: this thread's share of sum_b loss_b, already known to the caller -- get_loss is not called then;
    //  la_lanes_0_32: only lanes 0 and 32 of a wave hold a share, the wave's sum is one shuffle instead of six)
    // Jacobian rows are needed last but depend on nothing computed here: fetch them first (into LDS, not registers:
    // this body also runs inside the composite kernel, which is compiled for 80 registers)
    // (Js_lds: 6 x 16 floats of the caller's LDS -- the merged job kernel has none to spare, it lends a job's work area)
    float (*const Js)[16] = Js_lds;
    if (threadIdx.x < 96) Js[threadIdx.x >> 4][threadIdx.x & 15] = tc_jac[16 + threadIdx.x];
    float P[16];
    projection(K, H, W, n, f, P);
    for (int r = 0; r < 4; r++) {  // PF = proj @ opencv2blender
        P[4 * r + 1] = -P[4 * r + 1];
        P[4 * r + 2] = -P[4 * r + 2];
    }
    // work item = (view-link pair, row r of d/dTc <G, PF @ Tc @ lp> = PF^T @ G @ lp^T): four threads per pair
    const int r = threadIdx.x & 3;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    double la = 0.0;
    for (int i = threadIdx.x >> 2; i < B * L; i += blockDim.x >> 2) {
        // (the link pose is requested BEFORE the gradients: get_g may read them with atomic loads, behind which the
        //  compiler does not move an ordinary load -- a round trip of its own otherwise)
        float lp[16];
#pragma unroll
        for (int k = 0; k < 16; k++) lp[k] = link_poses[(size_t)i * 16 + k];
        float G[16];
        get_g(i, G);
        float M[4];
        for (int c = 0; c < 4; c++) {
            float s = 0.f;
            for (int k = 0; k < 4; k++) s = fmaf(P[4 * k + r], G[4 * k + c], s);
            M[c] = s;
        }
        for (int c = 0; c < 4; c++) {
            float s = 0.f;
            for (int k = 0; k < 4; k++) s = fmaf(M[k], lp[4 * c + k], s);
            acc[c] += (double)s;
        }
    }
    if (la_pre)
        la = *la_pre;
    else
        for (int i = threadIdx.x; i < B; i += blockDim.x) la += (double)get_loss(i);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // lanes with the same r hold the same row: butterfly over the lane bits above the row index, then the four waves
    // (the four columns in 5 shuffles instead of 16: a lane keeps half of its values at the first two levels and hands
    //  the other half to its partner -- the pairings, hence the sums, are the plain butterfly's, bit for bit; column
    //  c = 2 b32 + b16 of row r ends up in lane r + 16 b16 + 32 b32)
    {
        const bool b32 = lane & 32, b16 = lane & 16;
        const double a0 = (b32 ? acc[2] : acc[0]) + wave_xor<32>(b32 ? acc[0] : acc[2]);
        const double a1 = (b32 ? acc[3] : acc[1]) + wave_xor<32>(b32 ? acc[1] : acc[3]);
        double v = (b16 ? a1 : a0) + wave_xor<16>(b16 ? a0 : a1);
        v += wave_xor<8>(v);
        v += wave_xor<4>(v);
        if ((lane & 12) == 0) S[wave][4 * (lane & 3) + 2 * (b32 ? 1 : 0) + (b16 ? 1 : 0)] = v;
    }
    {
        const double s = la_lanes_0_32 ? la + wave_xor<32>(la) : wave_sum_f64(la);
        if (lane == 0) S[wave][16] = s;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        float rr;
        if (threadIdx.x < 6) {
            double g = 0.0;
            for (int k = 0; k < 16; k++)
                g += (((S[0][k] + S[1][k]) + S[2][k]) + S[3][k]) * (double)Js[threadIdx.x][k];
            rr = (float)g;
        } else {
            rr = (threadIdx.x == 6) ? (float)(((S[0][16] + S[1][16]) + S[2][16]) + S[3][16]) : (float)B;
        }
        red[threadIdx.x] = rr;
        if (red_lds) red_lds[threadIdx.x] = rr;
    }
}

__device__ __forceinline__ void pose_backward_block(const float* __restrict__ grad_mvp, const float* __restrict__ loss,
                                                    const float* __restrict__ K, const float* __restrict__ link_poses,
                                                    const float* __restrict__ tc_jac, int B, int L, int H, int W,
                                                    float n, float f, float* __restrict__ red, double (*S)[17]) {
    __shared__ float Js[6][16];
    pose_backward_block_t(
        [&](int i, float* G) {
            for (int k = 0; k < 16; k++) G[k] = grad_mvp[(size_t)i * 16 + k];
        },
        [&](int b) { return loss[b]; }, K, link_poses, tc_jac, B, L, H, W, n, f, red, S, nullptr, nullptr, false, Js);
}

// torch.optim.Adam with L2 weight decay on dof, gradient of the MEAN per-frame loss = red[0..5] / red[7].
// Call with >= 7 threads of one workgroup; contains a barrier.  `red` may point to LDS (written by this workgroup
// before a barrier) or to global memory.
struct AdamState {  // one thread's share of the optimiser state, fetched ahead of use
    float p, m, v;
    int t;
    float step_size, rsq_bc2;  // lr / (1 - b1^t) and sqrt(1 - b2^t): pose_adam_bias, computable as soon as t is known
    bool has_bias;
};
// The bias corrections depend on the step count alone: two powf and a square root that a caller waiting for memory
// anyway (the fused step's finish stage) takes off the end of its chain.  Same expressions as in pose_adam_apply.
__device__ __forceinline__ void pose_adam_bias(AdamState& st, float lr, float b1, float b2) {
    const float bc1 = 1.f - powf(b1, (float)st.t);
    const float bc2 = 1.f - powf(b2, (float)st.t);
    st.step_size = lr / bc1;
    st.rsq_bc2 = sqrtf(bc2);
    st.has_bias = true;
}
__device__ __forceinline__ AdamState pose_adam_fetch(const float* dof, const float* m, const float* v, const int* step) {
    AdamState st;
    st.p = st.m = st.v = 0.f;
    st.step_size = st.rsq_bc2 = 0.f;
    st.has_bias = false;
    st.t = step[0] + 1;
    if (threadIdx.x < 6) {
        st.p = dof[threadIdx.x];
        st.m = m[threadIdx.x];
        st.v = v[threadIdx.x];
    }
    return st;
}
__device__ __forceinline__ void pose_adam_apply(const AdamState& st, float* __restrict__ dof, float* __restrict__ m,
                                                float* __restrict__ v, int* __restrict__ step, const float* red, float lr,
                                                float b1, float b2, float eps, float wd, float* __restrict__ loss_out,
                                                float* __restrict__ grad_out) {
    const int i = threadIdx.x;
    const int t = st.t;
    const float nfr = red[7];
    // A reported failure (queue / accumulator overflow => NaN loss and gradient) must not destroy the calibration:
    // the parameters, the moments and the step counter stay as they are, only the NaN loss is published.
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 8; k++) ok = ok && (fabsf(red[k]) < 3.0e38f);
    if (i < 6 && ok) {
        float g = red[i] / nfr;
        if (grad_out) grad_out[i] = g;
        float p = st.p;
        g = g + wd * p;
        float mi = b1 * st.m + (1.f - b1) * g;
        float vi = b2 * st.v + (1.f - b2) * g * g;
        m[i] = mi;
        v[i] = vi;
        float step_size, rsq_bc2;
        if (st.has_bias) {
            step_size = st.step_size;
            rsq_bc2 = st.rsq_bc2;
        } else {
            const float bc1 = 1.f - powf(b1, (float)t);
            const float bc2 = 1.f - powf(b2, (float)t);
            step_size = lr / bc1;
            rsq_bc2 = sqrtf(bc2);
        }
        float denom = sqrtf(vi) / rsq_bc2 + eps;
        dof[i] = p - step_size * (mi / denom);
    }
    if (i < 6 && !ok && grad_out) grad_out[i] = __int_as_float(0x7fc00000);
    if (i == 6 && loss_out) loss_out[0] = ok ? red[6] / nfr : __int_as_float(0x7fc00000);
    __syncthreads();
    if (i == 0 && ok) step[0] = t;
}
__device__ __forceinline__ void pose_adam_block(float* __restrict__ dof, float* __restrict__ m, float* __restrict__ v,
                                                int* __restrict__ step, const float* __restrict__ red, float lr,
                                                float b1, float b2, float eps, float wd, float* __restrict__ loss_out,
                                                float* __restrict__ grad_out) {
    const AdamState st = pose_adam_fetch(dof, m, v, step);
    pose_adam_apply(st, dof, m, v, step, red, lr, b1, b2, eps, wd, loss_out, grad_out);
}

}  // namespace ehr
