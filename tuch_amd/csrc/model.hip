// tuch_contact_model: uploads the per-model constants once (faces, bit-packed
// geodesic mask, segment tables, region tables).  The only place the library
// allocates device memory; the hot calls never do.
#include "model.h"
#include <stdlib.h>
#include <string.h>
#include <vector>

extern "C" int tuch_geomask_words(int V);

namespace {

template <typename T>
int upload(T** dst, const T* src, size_t count)
{
    *dst = nullptr;
    if (count == 0) return TUCH_OK;
    if (hipMalloc((void**)dst, count * sizeof(T)) != hipSuccess) {
        tuch_set_error("tuch_contact_model_create: hipMalloc(%zu) failed", count * sizeof(T));
        return TUCH_ERR_HIP;
    }
    if (hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) {
        tuch_set_error("tuch_contact_model_create: hipMemcpy failed");
        return TUCH_ERR_HIP;
    }
    return TUCH_OK;
}

int* host_copy(const int32_t* src, size_t n)
{
    int* p = (int*)malloc(sizeof(int) * (n ? n : 1));
    if (n) memcpy(p, src, sizeof(int) * n);
    return p;
}

}  // namespace

extern "C" void tuch_contact_model_destroy(tuch_contact_model* m)
{
    if (!m) return;
    void* dev[] = {m->faces, m->mask_bits, m->seg_q_off, m->seg_q_vidx, m->seg_f_off, m->seg_faces,
                   m->cap_off, m->cap_vidx, m->region_off, m->region_vidx, m->pairs};
    for (void* p : dev)
        if (p) (void)hipFree(p);
    free(m->seg_q_off_host);
    free(m->seg_f_off_host);
    free(m->region_off_host);
    free(m);
}

extern "C" int tuch_contact_model_create(
    tuch_contact_model** out, int V, int F, const int32_t* faces,
    const uint8_t* geomask,   // host [V,V] bytes (geod > geothres) or NULL
    int num_segments, const int32_t* seg_q_off, const int32_t* seg_q_vidx,
    const int32_t* seg_f_off, const int32_t* seg_faces,
    int num_caps, const int32_t* cap_off, const int32_t* cap_vidx,
    int num_regions, const int32_t* region_off, const int32_t* region_vidx,
    int num_pairs, const int32_t* pairs)
{
    TUCH_REQUIRE(out && faces && V > 0 && F > 0, "tuch_contact_model_create: bad mesh arguments");
    TUCH_REQUIRE(num_segments >= 0 && num_caps >= 0 && num_regions >= 0 && num_pairs >= 0,
                 "tuch_contact_model_create: negative table size");
    TUCH_REQUIRE(num_segments == 0 || (seg_q_off && seg_q_vidx && seg_f_off && seg_faces),
                 "tuch_contact_model_create: segment tables missing");
    TUCH_REQUIRE(num_caps == 0 || (cap_off && cap_vidx), "tuch_contact_model_create: cap tables missing");
    TUCH_REQUIRE(num_pairs == 0 || (num_regions > 0 && region_off && region_vidx && pairs),
                 "tuch_contact_model_create: region tables missing");
    for (int i = 0; i < F * 3; ++i)
        TUCH_REQUIRE(faces[i] >= 0 && faces[i] < V, "tuch_contact_model_create: face index %d out of range",
                     faces[i]);
    tuch_contact_model* m = (tuch_contact_model*)calloc(1, sizeof(tuch_contact_model));
    m->V = V;
    m->F = F;
    (void)hipGetDevice(&m->device);
    int rc = upload(&m->faces, faces, (size_t)F * 3);
    if (rc == TUCH_OK && geomask) {
        // bit-pack on the host: bits[w][j], bit k = geomask[j][64 w + k]
        const int W = tuch_geomask_words(V);
        std::vector<uint64_t> bits((size_t)W * V, 0);
        for (int j = 0; j < V; ++j) {
            const uint8_t* row = geomask + (size_t)j * V;
            for (int i = 0; i < V; ++i)
                if (row[i]) bits[(size_t)(i >> 6) * V + j] |= (uint64_t)1 << (i & 63);
        }
        rc = upload(&m->mask_bits, bits.data(), bits.size());
    }
    if (rc == TUCH_OK && num_segments > 0) {
        m->num_segments = num_segments;
        m->num_caps = num_caps;
        m->seg_q_total = seg_q_off[num_segments];
        m->seg_f_total = seg_f_off[num_segments];
        m->seg_q_off_host = host_copy(seg_q_off, num_segments + 1);
        m->seg_f_off_host = host_copy(seg_f_off, num_segments + 1);
        for (int s = 0; s < num_segments; ++s) {
            const int n = seg_q_off[s + 1] - seg_q_off[s];
            if (n > m->seg_q_max) m->seg_q_max = n;
        }
        for (int i = 0; i < m->seg_f_total * 3 && rc == TUCH_OK; ++i)
            if (seg_faces[i] < 0 || seg_faces[i] >= V + num_caps) {
                tuch_set_error("tuch_contact_model_create: segment face index %d out of range", seg_faces[i]);
                rc = TUCH_ERR_ARG;
            }
        if (rc == TUCH_OK) rc = upload(&m->seg_q_off, seg_q_off, (size_t)num_segments + 1);
        if (rc == TUCH_OK) rc = upload(&m->seg_q_vidx, seg_q_vidx, (size_t)m->seg_q_total);
        if (rc == TUCH_OK) rc = upload(&m->seg_f_off, seg_f_off, (size_t)num_segments + 1);
        if (rc == TUCH_OK) rc = upload(&m->seg_faces, seg_faces, (size_t)m->seg_f_total * 3);
        if (rc == TUCH_OK && num_caps > 0) {
            rc = upload(&m->cap_off, cap_off, (size_t)num_caps + 1);
            if (rc == TUCH_OK) rc = upload(&m->cap_vidx, cap_vidx, (size_t)cap_off[num_caps]);
        }
    }
    if (rc == TUCH_OK && num_regions > 0) {
        m->num_regions = num_regions;
        m->num_pairs = num_pairs;
        m->region_off_host = host_copy(region_off, num_regions + 1);
        for (int r = 0; r < num_regions; ++r) {
            const int n = region_off[r + 1] - region_off[r];
            if (n > m->region_max) m->region_max = n;
        }
        rc = upload(&m->region_off, region_off, (size_t)num_regions + 1);
        if (rc == TUCH_OK) rc = upload(&m->region_vidx, region_vidx, (size_t)region_off[num_regions]);
        if (rc == TUCH_OK && num_pairs > 0) rc = upload(&m->pairs, pairs, (size_t)num_pairs * 2);
    }
    if (rc != TUCH_OK) {
        tuch_contact_model_destroy(m);
        *out = nullptr;
        return rc;
    }
    *out = m;
    return TUCH_OK;
}

extern "C" const uint64_t* tuch_contact_model_mask_bits(const tuch_contact_model* m)
{
    return m ? m->mask_bits : nullptr;
}

extern "C" const int32_t* tuch_contact_model_faces(const tuch_contact_model* m)
{
    return m ? m->faces : nullptr;
}

extern "C" int tuch_contact_model_info(const tuch_contact_model* m, int* V, int* F, int* num_segments,
                                       int* seg_q_total, int* num_pairs)
{
    TUCH_REQUIRE(m, "tuch_contact_model_info: null model");
    if (V) *V = m->V;
    if (F) *F = m->F;
    if (num_segments) *num_segments = m->num_segments;
    if (seg_q_total) *seg_q_total = m->seg_q_total;
    if (num_pairs) *num_pairs = m->num_pairs;
    return TUCH_OK;
}
