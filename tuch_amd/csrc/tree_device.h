// Device-side view of the cluster tree (cluster_tree.hip) shared by the winding and the
// nearest-vertex kernels.
#pragma once
#include "common.h"

struct TreeNode { int cap_off, cap_len, ex_off, ex_len, skip, c0, c1, nfaces; };
constexpr int kBoundsBlock = 256;

// Bounding slabs of the inner nodes of one body, bottom-up from the leaf slabs already in `bounds`
// ([B,N,2K] = K minima, then K maxima).  One workgroup per body, staged in LDS (N*8K bytes).
template <int K>
static __global__ __launch_bounds__(kBoundsBlock) void tree_inner_bounds_kernel(
    const TreeNode* __restrict__ nodes, int N, const int32_t* __restrict__ height_off,
    const int32_t* __restrict__ height_nodes, int num_heights, float* __restrict__ bounds,
    const int32_t* __restrict__ info = nullptr)      // [N][2] or nullptr: see below
{
    extern __shared__ float sb[];                       // N * 2K floats of slabs, then N * 2 ints of child indices
    constexpr int S = 2 * K;
    int* child = reinterpret_cast<int*>(sb + (size_t)N * S);
    float* out = bounds + (size_t)blockIdx.x * N * S;
    // one round of loads: the leaf slabs and every node's children (the per-level loop below then runs out of LDS; with
    // the child indices fetched from global memory inside it, every level was a dependent load latency)
    for (int i = height_off[0] + threadIdx.x; i < height_off[1]; i += kBoundsBlock) {
        const int node = height_nodes[i];
#pragma unroll
        for (int k = 0; k < S; ++k) sb[node * S + k] = out[node * S + k];
    }
    for (int node = threadIdx.x; node < N; node += kBoundsBlock) {
        child[2 * node] = nodes[node].c0;
        child[2 * node + 1] = nodes[node].c1;
    }
    __syncthreads();
    for (int h = 1; h < num_heights; ++h) {
        for (int i = height_off[h] + threadIdx.x; i < height_off[h + 1]; i += kBoundsBlock) {
            const int node = height_nodes[i];
            const float* a = sb + child[2 * node] * S;
            const float* c = sb + child[2 * node + 1] * S;
            float o[S];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                o[k] = fminf(a[k], c[k]);
                o[K + k] = fmaxf(a[K + k], c[K + k]);
            }
#pragma unroll
            for (int k = 0; k < S; ++k) {
                sb[node * S + k] = o[k];
                out[node * S + k] = o[k];
            }
        }
        __syncthreads();
    }
    // K = 4 (boxes): the two padding floats of every node carry two words of the caller's static node table, so that a
    // walk fetches a node's box and what it needs to know about the node in one load
    if (K == 4 && info)
        for (int node = threadIdx.x; node < N; node += kBoundsBlock) {
            out[node * S + 3] = __int_as_float(info[2 * node]);
            out[node * S + 7] = __int_as_float(info[2 * node + 1]);
        }
}

// LDS the kernel above stages for N nodes; above the default 64 KB per-launch limit the attribute has to be raised first
template <int K>
static inline size_t tree_inner_bounds_lds(int N)
{
    const size_t bytes = (size_t)N * (2 * K * sizeof(float) + 2 * sizeof(int));
    if (bytes > 64 * 1024) {
        static size_t raised = 0;                       // per instantiation
        if (bytes > raised &&
            hipFuncSetAttribute((const void*)tree_inner_bounds_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)bytes) == hipSuccess)
            raised = bytes;
    }
    return bytes;
}
