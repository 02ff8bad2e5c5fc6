// Device-resident constants of one body model (faces, packed geodesic mask,
// segment and region tables).  Created once, read-only in the hot calls.
#pragma once
#include "common.h"

struct tuch_contact_model {
    int device;
    int V, F;
    int32_t* faces;            // [F,3]
    uint64_t* mask_bits;       // [W][V] or nullptr
    // triangle strips over `faces` (built at create): stream of vertex ids with a per-element
    // sign: 0 = priming vertex (no triangle), +1/-1 = emit triangle (p-2, p-1, p) with that
    // orientation sign relative to the original face
    int strip_len, num_strips;
    int32_t* strip_vidx;       // [strip_len]
    float* strip_sign;         // [strip_len]
    // segments (tuch/utils/segmentation.py): CSR over segments
    int num_segments, num_caps, seg_q_total, seg_f_total;
    int32_t* seg_q_off;        // [S+1] into seg_q_vidx
    int32_t* seg_q_vidx;       // segment_vidx lists, concatenated
    int32_t* seg_f_off;        // [S+1] into seg_faces (in faces)
    int32_t* seg_faces;        // [seg_f_total,3], cap vertex c is index V + c
    int32_t* cap_off;          // [K+1] into cap_vidx
    int32_t* cap_vidx;         // ordered boundary loops, concatenated
    int num_seg_blocks;        // 128-query blocks over all segments
    int32_t* seg_blocks;       // [num_seg_blocks][2] = (segment, first query within the segment)
    int32_t* seg_of_q;         // [seg_q_total] segment of every entry of seg_q_vidx
    int* seg_q_off_host;       // host copies for grid sizing
    int* seg_f_off_host;
    int seg_q_max;
    // contact regions (ContactSigSMPL / classes): CSR over regions
    int num_regions, num_pairs, region_max;
    int32_t* region_off;       // [R+1]
    int32_t* region_vidx;
    int32_t* pairs;            // [P,2]
    // geodesic mask restricted to every region pair: rows = vertices of the first region,
    // bit k of a row = geomask[row vertex][k-th vertex of the second region]
    uint32_t* pair_mask;       // concatenated [n1][ceil(n2/32)] blocks, or nullptr
    int64_t* pair_mask_off;    // [P+1] word offsets
    int* region_off_host;
};
