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
#include "ehr_host.h"
#include "ehr_pose_core.h"

namespace ehr {

// state: [0..5] dTc/ddof written as 6 x 16 floats at jac; Tc at tc (16)
__global__ void __launch_bounds__(256) pose_forward_kernel(const float* __restrict__ dof, const float* __restrict__ K,
                                                           const float* __restrict__ link_poses, int B, int L, int H,
                                                           int W, float n, float f, float* __restrict__ mvp,
                                                           float* __restrict__ tc_jac, const int* __restrict__ step,
                                                           float* __restrict__ history, int history_rows) {
    __shared__ float Tc[16];
    if (threadIdx.x == 0) {
        D6 T[16];
        se3_exp_dual<6>(dof, 1e-4f, T);
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
        float C[16];
        mvp_from_pose(Tc, P, link_poses + (size_t)i * 16, C);
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
    __shared__ double S[4][17];
    pose_backward_block(grad_mvp, loss, K, link_poses, tc_jac, B, L, H, W, n, f, red, S);
}

// Adam on dof with the gradient of the MEAN per-frame loss: g = red[0..5] / red[7].  adam = {lr, b1, b2, eps, wd}.
// state: m[6], v[6]; step counter incremented here.  loss_out = red[6] / red[7].
__global__ void pose_adam_kernel(float* __restrict__ dof, float* __restrict__ m, float* __restrict__ v,
                                 int* __restrict__ step, const float* __restrict__ red, float lr, float b1, float b2,
                                 float eps, float wd, float* __restrict__ loss_out, float* __restrict__ grad_out) {
    pose_adam_block(dof, m, v, step, red, lr, b1, b2, eps, wd, loss_out, grad_out);
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
