// Harness that drives the reference's OWN opq sources, compiled in place from
// /root/reference/opq/src (see oracle/Makefile: target _ref).  Test infrastructure only.
// No reference source text lives in this file: it includes the reference header and is linked
// with the reference's IVFOPQ.cpp.  Output: oracle/_ref/libref_opq.so (git-ignored).
//
// The class keeps its tables private and Query() iterates m_ivfSize, which only LoadIndex fills
// (SURVEY.md 3.5), so the harness opens the class up with the usual test trick and fills
// m_ivfSize from the in-memory lists before querying.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <queue>
#include <sstream>
#include <string>
#include <vector>
#include <limits.h>
#include <sys/time.h>
#include <unistd.h>
#include <fcntl.h>

#define private public
#include "IVFOPQ.h"
#undef private
#include "common.h"

namespace {
// Add() printf()s every code and Query() prints tables: silence fd 1 around reference calls.
struct Quiet {
    int saved;
    Quiet() {
        fflush(stdout); std::cout.flush();
        saved = dup(1);
        int nul = open("/dev/null", O_WRONLY);
        dup2(nul, 1); close(nul);
    }
    ~Quiet() { fflush(stdout); std::cout.flush(); dup2(saved, 1); close(saved); }
};
}  // namespace

extern "C" {

__attribute__((visibility("default"))) void *ref_opq_new(const char *model_file, int max_index_num)
{
    Quiet q;
    IVFOPQ *h = new IVFOPQ(max_index_num);
    if (h->LoadModel(model_file) != 1) { delete h; return nullptr; }
    return h;
}

__attribute__((visibility("default"))) void ref_opq_delete(void *p) { delete (IVFOPQ *)p; }

__attribute__((visibility("default"))) void ref_opq_dims(void *p, int *D, int *coarseK, int *M, int *K)
{
    IVFOPQ *h = (IVFOPQ *)p;
    *D = h->m_featDim; *coarseK = h->m_coarseK; *M = h->m_pq_m; *K = h->m_pq_k;
}

// LoadSingleFeatFile: raw fp32 rows + reorder().  Returns the row count, copies min(rows,cap).
__attribute__((visibility("default"))) int ref_opq_load_feat(void *p, const char *file, float *out, int cap_rows)
{
    IVFOPQ *h = (IVFOPQ *)p;
    float **feat = nullptr; int n = 0;
    { Quiet q; h->LoadSingleFeatFile(file, feat, n); }
    if (n > 0) {
        int m = std::min(n, cap_rows);
        memcpy(out, feat[0], sizeof(float) * (size_t)m * h->m_featDim);
        Delete2DArray(feat);
    }
    return n;
}

// IndexDatabase over a list of feature files (one "video" per file).
__attribute__((visibility("default"))) int ref_opq_index(void *p, const char **files, int nfiles)
{
    IVFOPQ *h = (IVFOPQ *)p;
    std::vector<std::string> v(files, files + nfiles);
    Quiet q;
    h->IndexDatabase(v);
    return h->m_imgNum;
}

// Flatten the inverted lists in list order.  Returns the total entry count.
__attribute__((visibility("default"))) int64_t ref_opq_dump(void *p, int64_t *list_off /*[coarseK+1]*/, int32_t *video_id,
                                                  uint8_t *codes, int64_t cap)
{
    IVFOPQ *h = (IVFOPQ *)p;
    int64_t tot = 0;
    for (int l = 0; l < h->m_coarseK; ++l) {
        if (list_off) list_off[l] = tot;
        for (size_t j = 0; j < h->m_ivfList[l].size(); ++j) {
            if (tot < cap) {
                if (video_id) video_id[tot] = h->m_ivfList[l][j].videoId;
                if (codes) memcpy(codes + tot * h->m_pq_m, h->m_ivfList[l][j].PQindex, (size_t)h->m_pq_m);
            }
            ++tot;
        }
    }
    if (list_off) list_off[h->m_coarseK] = tot;
    return tot;
}

// QueryThrehold (== Query without the debug prints).  match_score is [frames][imgNum].
__attribute__((visibility("default"))) int ref_opq_query(void *p, const char *file, int nk, float *match_score, int64_t cap_floats,
                                               int *frames, int *img_num)
{
    IVFOPQ *h = (IVFOPQ *)p;
    if (h->m_ivfSize == NULL) h->m_ivfSize = new int[h->m_coarseK];
    for (int l = 0; l < h->m_coarseK; ++l) h->m_ivfSize[l] = (int)h->m_ivfList[l].size();
    std::vector<std::vector<float> > score;
    { Quiet q; h->QueryThrehold(file, score, nk); }
    *frames = (int)score.size();
    *img_num = h->m_imgNum;
    int64_t w = 0;
    for (size_t f = 0; f < score.size(); ++f)
        for (size_t v = 0; v < score[f].size(); ++v)
            if (w < cap_floats) match_score[w++] = score[f][v];
    return 0;
}

__attribute__((visibility("default"))) int ref_opq_save_index(void *p, const char *dir)
{
    Quiet q;
    ((IVFOPQ *)p)->SaveIndex(dir);
    return 0;
}

// get_sort_results (common.h): k smallest (score, index) ascending.
__attribute__((visibility("default"))) void ref_sort_results(const float *score, int n, int k, float *out_d, uint32_t *out_id)
{
    std::vector<float> s(score, score + n);
    std::vector<std::pair<float, uint> > r = get_sort_results(s, k);
    for (int i = 0; i < k; ++i) { out_d[i] = r[i].first; out_id[i] = r[i].second; }
}

}  // extern "C"
