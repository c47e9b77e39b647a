// ehr_pose.hip -- the two ends of the optimisation step that surround the renderer, as single-workgroup kernels:
//
//   pose_forward : dof[6] -> Tc_c2b = se3_exp_map(dof)  (restates /root/reference/easyhec/utils/pytorch3d_se3.py:12-41,
//                  :46-130, :218-245 incl. the squared-angle clamp) -> MVP[b,l] = proj @ (opencv2blender @ (Tc_c2b @
//                  link_pose[b,l]))  (rb_solver.py:52,63; nvdiffrast_renderer.py:33-37; nvdiffrast_utils.py:5-11),
//                  plus d Tc_c2b / d dof_i by forward-mode differentiation of the same formulas, and the
//                  history_ops row of rb_solver.py:50-51.
//   pose_backward: d loss_b / d MVP[b,l] -> d (sum_b loss_b) / d dof, the loss sum and the frame count (the 8 floats
//                  the data-parallel all-reduce exchanges).
//   pose_adam    : torch.optim.Adam step with L2 weight decay on dof (solver/build.py:12-29; defaults.py:138).
//
// In the reference this is ~190 tiny torch kernels + autograd per step; here it is 3 launches.
#include "ehr_device.h"
#include "ehr_host.h"

namespace ehr {

// ---- forward-mode scalar with 6 partials -------------------------------------------------------------------------
struct D6 {
    float v;
    float d[6];
};
__device__ __forceinline__ D6 dconst(float c) {
    D6 r;
    r.v = c;
#pragma unroll
    for (int i = 0; i < 6; i++) r.d[i] = 0.f;
    return r;
}
__device__ __forceinline__ D6 dvar(float c, int i) {
    D6 r = dconst(c);
    r.d[i] = 1.f;
    return r;
}
__device__ __forceinline__ D6 operator+(const D6& a, const D6& b) {
    D6 r;
    r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < 6; i++) r.d[i] = a.d[i] + b.d[i];
    return r;
}
__device__ __forceinline__ D6 operator-(const D6& a, const D6& b) {
    D6 r;
    r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < 6; i++) r.d[i] = a.d[i] - b.d[i];
    return r;
}
__device__ __forceinline__ D6 operator*(const D6& a, const D6& b) {
    D6 r;
    r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < 6; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
__device__ __forceinline__ D6 operator/(const D6& a, const D6& b) {
    D6 r;
    float inv = 1.f / b.v;
    r.v = a.v * inv;
#pragma unroll
    for (int i = 0; i < 6; i++) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
__device__ __forceinline__ D6 dneg(const D6& a) { return dconst(0.f) - a; }
__device__ __forceinline__ D6 dsqrt(const D6& a) {
    D6 r;
    r.v = sqrtf(a.v);
    float k = 0.5f / r.v;
#pragma unroll
    for (int i = 0; i < 6; i++) r.d[i] = a.d[i] * k;
    return r;
}
__device__ __forceinline__ D6 dsin(const D6& a) {
    D6 r;
    r.v = sinf(a.v);
    float c = cosf(a.v);
#pragma unroll
    for (int i = 0; i < 6; i++) r.d[i] = a.d[i] * c;
    return r;
}
__device__ __forceinline__ D6 dcos(const D6& a) {
    D6 r;
    r.v = cosf(a.v);
    float s = -sinf(a.v);
#pragma unroll
    for (int i = 0; i < 6; i++) r.d[i] = a.d[i] * s;
    return r;
}
// torch.clamp(x, min=eps): value max(x, eps), gradient passes only where x >= eps
__device__ __forceinline__ D6 dclamp_min(const D6& a, float eps) { return (a.v >= eps) ? a : dconst(eps); }

// Tc (row-major 4x4, the usual [[R, t], [0, 1]]) and its 6 partials from dof = [log_translation, log_rotation]
__device__ void se3_exp_dual(const float* dof, float eps, D6 T[16]) {
    D6 u0 = dvar(dof[0], 0), u1 = dvar(dof[1], 1), u2 = dvar(dof[2], 2);
    D6 w0 = dvar(dof[3], 3), w1 = dvar(dof[4], 4), w2 = dvar(dof[5], 5);
    D6 nrms = w0 * w0 + w1 * w1 + w2 * w2;
    D6 th = dsqrt(dclamp_min(nrms, eps));
    D6 one = dconst(1.f), zero = dconst(0.f);
    D6 inv = one / th;
    D6 fac1 = inv * dsin(th);
    D6 fac2 = inv * inv * (one - dcos(th));
    // hat(w) and its square
    D6 K[9] = {zero, dneg(w2), w1, w2, zero, dneg(w0), dneg(w1), w0, zero};
    D6 K2[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) K2[3 * r + c] = K[3 * r] * K[c] + K[3 * r + 1] * K[3 + c] + K[3 * r + 2] * K[6 + c];
    D6 fv1 = (one - dcos(th)) / (th * th);
    D6 fv2 = (th - dsin(th)) / (th * th * th);
    D6 u[3] = {u0, u1, u2};
    for (int r = 0; r < 3; r++) {
        D6 t = zero;
        for (int c = 0; c < 3; c++) {
            D6 I = dconst(r == c ? 1.f : 0.f);
            T[4 * r + c] = fac1 * K[3 * r + c] + fac2 * K2[3 * r + c] + I;
            D6 V = I + K[3 * r + c] * fv1 + K2[3 * r + c] * fv2;
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

// state: [0..5] dTc/ddof written as 6 x 16 floats at jac; Tc at tc (16)
__global__ void __launch_bounds__(256) pose_forward_kernel(const float* __restrict__ dof, const float* __restrict__ K,
                                                           const float* __restrict__ link_poses, int B, int L, int H,
                                                           int W, float n, float f, float* __restrict__ mvp,
                                                           float* __restrict__ tc_jac, const int* __restrict__ step,
                                                           float* __restrict__ history, int history_rows) {
    __shared__ float Tc[16];
    if (threadIdx.x == 0) {
        D6 T[16];
        se3_exp_dual(dof, 1e-4f, T);
        for (int i = 0; i < 16; i++) {
            Tc[i] = T[i].v;
            tc_jac[i] = T[i].v;
            for (int k = 0; k < 6; k++) tc_jac[16 * (k + 1) + i] = T[i].d[k];
        }
        if (history && step) {
            int row = step[0];
            if (row >= 0 && row < history_rows)
                for (int k = 0; k < 6; k++) history[6 * row + k] = dof[k];
        }
    }
    __syncthreads();
    float P[16];
    projection(K, H, W, n, f, P);
    for (int i = threadIdx.x; i < B * L; i += blockDim.x) {
        const float* lp = link_poses + (size_t)i * 16;
        float A[16], C[16];
        mat4_mul(Tc, lp, A);  // Tc_c2l = Tc_c2b @ link_pose            (rb_solver.py:63)
        // opencv2blender = diag(1,-1,-1,1)                              (nvdiffrast_renderer.py:35)
        for (int c = 0; c < 4; c++) {
            A[4 + c] = -A[4 + c];
            A[8 + c] = -A[8 + c];
        }
        mat4_mul(P, A, C);    // proj @ pose                             (nvdiffrast_renderer.py:37)
        for (int k = 0; k < 16; k++) mvp[(size_t)i * 16 + k] = C[k];
    }
}

// red[0..5] = d(sum_b loss_b)/d dof, red[6] = sum_b loss_b, red[7] = B.   Single workgroup, fixed-order reductions.
__global__ void __launch_bounds__(256) pose_backward_kernel(const float* __restrict__ grad_mvp,
                                                            const float* __restrict__ loss,
                                                            const float* __restrict__ K,
                                                            const float* __restrict__ link_poses,
                                                            const float* __restrict__ tc_jac, int B, int L, int H, int W,
                                                            float n, float f, float* __restrict__ red) {
    __shared__ double S[256][16];
    __shared__ double lsum[256];
    float P[16];
    projection(K, H, W, n, f, P);
    // PF = proj @ opencv2blender
    for (int r = 0; r < 4; r++) {
        P[4 * r + 1] = -P[4 * r + 1];
        P[4 * r + 2] = -P[4 * r + 2];
    }
    double acc[16];
    for (int k = 0; k < 16; k++) acc[k] = 0.0;
    double la = 0.0;
    for (int i = threadIdx.x; i < B * L; i += blockDim.x) {
        const float* G = grad_mvp + (size_t)i * 16;
        const float* lp = link_poses + (size_t)i * 16;
        // d/dTc of <G, PF @ Tc @ lp>  =  PF^T @ G @ lp^T
        float M[16];
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) {
                float s = 0.f;
                for (int k = 0; k < 4; k++) s = fmaf(P[4 * k + r], G[4 * k + c], s);
                M[4 * r + c] = s;
            }
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) {
                float s = 0.f;
                for (int k = 0; k < 4; k++) s = fmaf(M[4 * r + k], lp[4 * c + k], s);
                acc[4 * r + c] += (double)s;
            }
    }
    for (int i = threadIdx.x; i < B; i += blockDim.x) la += (double)loss[i];
    for (int k = 0; k < 16; k++) S[threadIdx.x][k] = acc[k];
    lsum[threadIdx.x] = la;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            for (int k = 0; k < 16; k++) S[threadIdx.x][k] += S[threadIdx.x + o][k];
            lsum[threadIdx.x] += lsum[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x < 6) {
        const float* J = tc_jac + 16 * (threadIdx.x + 1);
        double g = 0.0;
        for (int k = 0; k < 16; k++) g += S[0][k] * (double)J[k];
        red[threadIdx.x] = (float)g;
    }
    if (threadIdx.x == 6) red[6] = (float)lsum[0];
    if (threadIdx.x == 7) red[7] = (float)B;
}

// Adam on dof with the gradient of the MEAN per-frame loss: g = red[0..5] / red[7].  adam = {lr, b1, b2, eps, wd}.
// state: m[6], v[6]; step counter incremented here.  loss_out = red[6] / red[7].
__global__ void pose_adam_kernel(float* __restrict__ dof, float* __restrict__ m, float* __restrict__ v,
                                 int* __restrict__ step, const float* __restrict__ red, float lr, float b1, float b2,
                                 float eps, float wd, float* __restrict__ loss_out, float* __restrict__ grad_out) {
    int i = threadIdx.x;
    int t = step[0] + 1;
    float nfr = red[7];
    if (i < 6) {
        float g = red[i] / nfr;
        if (grad_out) grad_out[i] = g;
        float p = dof[i];
        g = g + wd * p;
        float mi = b1 * m[i] + (1.f - b1) * g;
        float vi = b2 * v[i] + (1.f - b2) * g * g;
        m[i] = mi;
        v[i] = vi;
        float bc1 = 1.f - powf(b1, (float)t);
        float bc2 = 1.f - powf(b2, (float)t);
        float step_size = lr / bc1;
        float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        dof[i] = p - step_size * (mi / denom);
    }
    if (i == 6 && loss_out) loss_out[0] = red[6] / nfr;
    __syncthreads();
    if (i == 0) step[0] = t;
}

}  // namespace ehr

using namespace ehr;

extern "C" {

int ehr_pose_forward(const float* dof, const float* K, const float* link_poses, int B, int L, int H, int W, float n,
                     float f, float* mvp, float* tc_jac, const int32_t* step, float* history, int history_rows,
                     void* stream) {
    if (!dof || !K || !link_poses || !mvp || !tc_jac) return fail(EHR_ERR_INVALID, "ehr_pose_forward: NULL tensor");
    if (B <= 0 || L <= 0) return fail(EHR_ERR_INVALID, "ehr_pose_forward: bad sizes");
    pose_forward_kernel<<<1, 256, 0, (hipStream_t)stream>>>(dof, K, link_poses, B, L, H, W, n, f, mvp, tc_jac, step,
                                                           history, history_rows);
    EHR_LAUNCH_CHECK();
    return EHR_OK;
}

int ehr_pose_backward(const float* grad_mvp, const float* loss, const float* K, const float* link_poses,
                      const float* tc_jac, int B, int L, int H, int W, float n, float f, float* red, void* stream) {
    if (!grad_mvp || !loss || !K || !link_poses || !tc_jac || !red)
        return fail(EHR_ERR_INVALID, "ehr_pose_backward: NULL tensor");
    pose_backward_kernel<<<1, 256, 0, (hipStream_t)stream>>>(grad_mvp, loss, K, link_poses, tc_jac, B, L, H, W, n, f, red);
    EHR_LAUNCH_CHECK();
    return EHR_OK;
}

int ehr_pose_adam(float* dof, float* m, float* v, int32_t* step, const float* red, float lr, float beta1, float beta2,
                  float eps, float weight_decay, float* loss_out, float* grad_out, void* stream) {
    if (!dof || !m || !v || !step || !red) return fail(EHR_ERR_INVALID, "ehr_pose_adam: NULL tensor");
    pose_adam_kernel<<<1, 64, 0, (hipStream_t)stream>>>(dof, m, v, step, red, lr, beta1, beta2, eps, weight_decay,
                                                       loss_out, grad_out);
    EHR_LAUNCH_CHECK();
    return EHR_OK;
}

}  // extern "C"
