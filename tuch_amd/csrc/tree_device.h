// Device-side view of the cluster tree (cluster_tree.hip) shared by the winding and the
// nearest-vertex kernels.
#pragma once
#include "common.h"

struct TreeNode { int cap_off, cap_len, ex_off, ex_len, skip, c0, c1, nfaces; };
constexpr int kBoundsBlock = 256;

// Boxes of the inner nodes of one body, bottom-up from the leaf boxes already in `bounds`
// ([B,N,8] = min xyz, -, max xyz, -).  One workgroup per body, boxes staged in LDS (N*32 bytes).
static __global__ __launch_bounds__(kBoundsBlock) void tree_inner_bounds_kernel(
    const TreeNode* __restrict__ nodes, int N, const int32_t* __restrict__ height_off,
    const int32_t* __restrict__ height_nodes, int num_heights, float* __restrict__ bounds)
{
    extern __shared__ float sb[];
    float* out = bounds + (size_t)blockIdx.x * N * 8;
    for (int i = height_off[0] + threadIdx.x; i < height_off[1]; i += kBoundsBlock) {
        const int node = height_nodes[i];
#pragma unroll
        for (int k = 0; k < 8; ++k) sb[node * 8 + k] = out[node * 8 + k];
    }
    __syncthreads();
    for (int h = 1; h < num_heights; ++h) {
        for (int i = height_off[h] + threadIdx.x; i < height_off[h + 1]; i += kBoundsBlock) {
            const int node = height_nodes[i];
            const float* a = sb + nodes[node].c0 * 8;
            const float* c = sb + nodes[node].c1 * 8;
            float o[8];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                o[k] = fminf(a[k], c[k]);
                o[4 + k] = fmaxf(a[4 + k], c[4 + k]);
            }
            o[3] = o[7] = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                sb[node * 8 + k] = o[k];
                out[node * 8 + k] = o[k];
            }
        }
        __syncthreads();
    }
}
