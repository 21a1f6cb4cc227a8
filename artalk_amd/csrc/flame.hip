// FLAME linear blend skinning: the first consumer of the 106-d codes (reference app/flame_model/FLAME.py:117-142 ->
// app/flame_model/lbs.py:142-233), SURVEY.md section 8f rank 3.  106-d motion + 300-d shape -> 5023 x 3 vertices per frame.
//   v_shaped = v_template + shapedirs . betas                 (lbs.py:185, blend_shapes :262)  -> fp32 MFMA GEMM, K = 400
//   J        = J_regressor . v_shaped                          (lbs.py:189, vertices2joints :241) -> flame_joints_kernel
//   R_j      = rodrigues(pose_j); pose_feature = R_1.. - I     (lbs.py:195-198, batch_rodrigues :266) -> flame_pose_kernel
//   v_posed  = v_shaped + pose_feature . posedirs              (lbs.py:200-212)                  -> fp32 MFMA GEMM, K = 36
//   A_j      = rigid transform chain relative to the rest pose (lbs.py:214, batch_rigid_transform :313)
//   verts    = (sum_j w_vj A_j) [v_posed, 1]                   (lbs.py:216-231)                  -> flame_skin_kernel
#include "common.h"

namespace artalk {

constexpr int FJ = 5;   // joints of FLAME: global, neck, jaw, left eye, right eye

// one thread per (frame, joint): axis-angle -> rotation matrix; joints >= 1 also emit R - I into the pose-feature row
__global__ void flame_pose_kernel(const float* __restrict__ pose /*[T][FJ*3]*/, float* __restrict__ rot /*[T][FJ][9]*/,
                                  float* __restrict__ feat /*[T][ldf]*/, int T, int ldf) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * FJ) return;
    const int t = idx / FJ, j = idx % FJ;
    const float* p = pose + (long)t * FJ * 3 + j * 3;
    const float x = p[0], y = p[1], z = p[2];
    const float ex = x + 1e-8f, ey = y + 1e-8f, ez = z + 1e-8f;
    const float angle = sqrtf(ex * ex + ey * ey + ez * ez);          // torch.norm(rot_vecs + 1e-8)
    const float rx = x / angle, ry = y / angle, rz = z / angle;
    const float s = sinf(angle), c = cosf(angle);
    const float K[9] = {0.f, -rz, ry, rz, 0.f, -rx, -ry, rx, 0.f};
    float KK[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) KK[a * 3 + b] = K[a * 3] * K[b] + K[a * 3 + 1] * K[3 + b] + K[a * 3 + 2] * K[6 + b];
    float R[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) R[e] = ((e % 4 == 0) ? 1.f : 0.f) + s * K[e] + (1.f - c) * KK[e];
#pragma unroll
    for (int e = 0; e < 9; ++e) rot[((long)t * FJ + j) * 9 + e] = R[e];
    if (j >= 1)
#pragma unroll
        for (int e = 0; e < 9; ++e) feat[(long)t * ldf + (j - 1) * 9 + e] = R[e] - ((e % 4 == 0) ? 1.f : 0.f);
}

// one workgroup per frame: J[t][j][k] = sum_v Jreg[j][v] * v_shaped[t][v][k]
__global__ __launch_bounds__(256) void flame_joints_kernel(const float* __restrict__ vs /*[T][V*3]*/, const float* __restrict__ jreg /*[FJ][V]*/,
                                                           float* __restrict__ J /*[T][FJ][3]*/, int V) {
    __shared__ float red[4][FJ * 3];
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float acc[FJ * 3];
#pragma unroll
    for (int e = 0; e < FJ * 3; ++e) acc[e] = 0.f;
    const float* v = vs + (long)t * V * 3;
    for (int i = tid; i < V; i += 256) {
        const float x = v[i * 3], y = v[i * 3 + 1], z = v[i * 3 + 2];
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
            const float w = jreg[(long)j * V + i];
            acc[j * 3] = fmaf(w, x, acc[j * 3]); acc[j * 3 + 1] = fmaf(w, y, acc[j * 3 + 1]); acc[j * 3 + 2] = fmaf(w, z, acc[j * 3 + 2]);
        }
    }
#pragma unroll
    for (int e = 0; e < FJ * 3; ++e) {
        const float s = wave_sum(acc[e]);
        if (lane == 0) red[wv][e] = s;
    }
    __syncthreads();
    if (tid < FJ * 3) J[(long)t * FJ * 3 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// grid (ceil(V/256), T): rigid transform chain per frame (recomputed per block: 5 joints), then skin 256 vertices
__global__ __launch_bounds__(256) void flame_skin_kernel(const float* __restrict__ vposed /*[T][V*3]*/, const float* __restrict__ rot,
                                                         const float* __restrict__ J, const int* __restrict__ parents,
                                                         const float* __restrict__ weights /*[V][FJ]*/, float* __restrict__ out,
                                                         int V, float scale) {
    __shared__ float A[FJ][12];      // rows 0..2 of the relative transforms, row-major 3x4
    const int t = blockIdx.y;
    if (threadIdx.x == 0) {
        float G[FJ][12];
        const float* R = rot + (long)t * FJ * 9;
        const float* Jt = J + (long)t * FJ * 3;
        for (int j = 0; j < FJ; ++j) {
            const int par = parents[j];
            float rel[3];
            for (int k = 0; k < 3; ++k) rel[k] = (j == 0) ? Jt[k] : Jt[j * 3 + k] - Jt[par * 3 + k];
            float Tm[12];
            for (int a = 0; a < 3; ++a) { for (int b = 0; b < 3; ++b) Tm[a * 4 + b] = R[j * 9 + a * 3 + b]; Tm[a * 4 + 3] = rel[a]; }
            if (j == 0) {
                for (int e = 0; e < 12; ++e) G[0][e] = Tm[e];
            } else {   // G_j = G_parent * T_j   (bottom rows are [0 0 0 1])
                for (int a = 0; a < 3; ++a) {
                    for (int b = 0; b < 4; ++b) {
                        float s = 0.f;
                        for (int k = 0; k < 3; ++k) s += G[par][a * 4 + k] * Tm[k * 4 + b];
                        if (b == 3) s += G[par][a * 4 + 3];
                        G[j][a * 4 + b] = s;
                    }
                }
            }
        }
        for (int j = 0; j < FJ; ++j)       // A_j = G_j with translation G_j.t - G_j.R * J_j  (lbs.py:371-372)
            for (int a = 0; a < 3; ++a) {
                float s = 0.f;
                for (int k = 0; k < 3; ++k) { A[j][a * 4 + k] = G[j][a * 4 + k]; s += G[j][a * 4 + k] * Jt[j * 3 + k]; }
                A[j][a * 4 + 3] = G[j][a * 4 + 3] - s;
            }
    }
    __syncthreads();
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    float Tm[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) Tm[e] = 0.f;
#pragma unroll
    for (int j = 0; j < FJ; ++j) {
        const float w = weights[(long)v * FJ + j];
#pragma unroll
        for (int e = 0; e < 12; ++e) Tm[e] = fmaf(w, A[j][e], Tm[e]);
    }
    const float* p = vposed + ((long)t * V + v) * 3;
    const float x = p[0], y = p[1], z = p[2];
    float* o = out + ((long)t * V + v) * 3;
#pragma unroll
    for (int a = 0; a < 3; ++a) o[a] = (Tm[a * 4] * x + Tm[a * 4 + 1] * y + Tm[a * 4 + 2] * z + Tm[a * 4 + 3]) * scale;
}

void launch_flame_pose(const float* pose, float* rot, float* feat, int T, int ldf, hipStream_t s) {
    ARTALK_LAUNCH(flame_pose_kernel, dim3((T * FJ + 127) / 128), dim3(128), 0, s, pose, rot, feat, T, ldf);
}
void launch_flame_joints(const float* vs, const float* jreg, float* J, int T, int V, hipStream_t s) {
    ARTALK_LAUNCH(flame_joints_kernel, dim3(T), dim3(256), 0, s, vs, jreg, J, V);
}
void launch_flame_skin(const float* vposed, const float* rot, const float* J, const int* parents, const float* weights, float* out,
                       int T, int V, float scale, hipStream_t s) {
    ARTALK_LAUNCH(flame_skin_kernel, dim3((V + 255) / 256, T), dim3(256), 0, s, vposed, rot, J, parents, weights, out, V, scale);
}

}  // namespace artalk
