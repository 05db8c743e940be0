// IVFOPQ.h -- drop-in mirror of the reference's IVFOPQ class (opq/src/IVFOPQ.h:31-69) whose hot loops
// run on the MI355X through the C ABI (include/cvtmi.h).  Same public surface, same argument meaning,
// same return conventions; the private tables of the reference are replaced by one cvtmi_opq_t handle
// (model + code index resident in HBM).
//
// Deliberate differences, all documented in SURVEY.md 3.5 / DESIGN.md:
//   - LoadIndex() really reads what SaveIndex() writes (the reference reads sizeof(IVFelem) per entry
//     and cannot round-trip its own files);
//   - Query() works right after IndexDatabase() (the reference needs m_ivfSize, which only LoadIndex fills);
//   - Add() does not printf every code; nothing leaks.
#ifndef CVT_AMD_IVFOPQ_H
#define CVT_AMD_IVFOPQ_H

#include <string>
#include <vector>

typedef unsigned char uchar;
const int max_path = 260;
struct ImgNameStruct {
    char ptr[max_path];
};

struct cvtmi_opq_s;
struct cvtmi_comm_s;

class IVFOPQ {
public:
    IVFOPQ();
    IVFOPQ(int maxIndexNum);
    ~IVFOPQ();
    int LoadModel(std::string modelFile);                         // 1 ok / 0 failure (IVFOPQ.cpp:64-102)
    void LoadIndex(std::string srcFile);                          // exit(0) on open failure (:464-472)
    void SaveIndex(std::string desDir);                           // layout of IVFOPQ.cpp:541-580
    void Add(float **m_ppFeat, const int m_frameNum);             // rows already rotated, row 0 = contiguous block
    void IndexDatabase(std::vector<std::string> featFiles);
    void Query(std::string featFile, std::vector<std::vector<float> > &matchScore, int nk = 3);
    void QueryThrehold(std::string featFile, std::vector<std::vector<float> > &matchScore, int nk = 3);
    void LoadSingleFeatFile(std::string srcFile, float **&m_ppFeat, int &m_frameNum);  // raw fp32 rows + rotation

    ImgNameStruct *m_imgLocation;

    // additions (not in the reference): shape accessors and the batched per-vector search of the north star
    int dim() const { return m_featDim; }
    int numImages() const { return m_imgNum; }
    long long numEntries() const;
    // k smallest (ADC distance, entry id) per query over every entry; needs coarseK == 1.  q is RAW (un-rotated).
    int SearchTopK(const float *q, int nq, int k, float *dist, long long *ids);
    std::string lastError() const;
    // row-sharded operation (one process per GPU): this object holds the row block that starts at global
    // row id_base; with a communicator set (cvtmi_comm_t, include/cvtmi.h) SearchTopK returns the GLOBAL top k on every
    // rank -- local scan, one RCCL all-gather of the per-shard lists, merge.  comm == NULL: back to single-GPU.
    void SetShard(cvtmi_comm_s *comm, long long id_base);
    // rotate + encode + append n RAW rows, one entry per row (the per-vector form of IndexDatabase); 1 ok / 0 failure
    int AddRows(const float *raw, int n);
    // ONE process, ndev GPUs (what a single-process caller like multi_frame_index_test.cpp:32-91 needs to scale): call after
    // LoadModel and before the first AddRows.  Device d keeps rows [d * cap, (d + 1) * cap), cap = ceil(expected_rows / ndev)
    // (the last device takes whatever comes beyond), in insertion order, so entry ids equal those of a single-GPU index;
    // SearchTopK scans every block side by side and merges after one grouped RCCL all-gather (cvtmi_comm_create_all +
    // cvtmi_opq_search_sharded_all).  AddRows / SearchTopK / numEntries work in this mode, the per-video calls do not.
    // 1 ok / 0 failure (fewer devices than asked for, RCCL not loadable, entries already present).
    int SetDevices(int ndev, long long expected_rows);
    // A dense D x D rotation (row-major fp32, y = R x: the fp32 MFMA GEMM of the library) INSTEAD of the model's reorder_ permutation --
    // e.g. one learned by cvtmi_opq_learn_rotation / TrainPQ::LearnRotation; the model file has no slot for it, so it lives beside the
    // model as raw fp32 [D][D] (LoadRotation).  Call after LoadModel and before anything is indexed.  1 ok / 0 failure.
    int SetRotation(const float *R);
    int LoadRotation(std::string rotationFile);
    int numDevices() const { return (int)m_hs.size() > 1 ? (int)m_hs.size() : 1; }

private:
    void init();
    bool ensureHandle();
    void queryImpl(const std::string &featFile, std::vector<std::vector<float> > &matchScore, int nk);

    cvtmi_opq_s *m_h;                     // single-GPU handle (= device 0's in multi-device mode)
    cvtmi_comm_s *m_comm;
    long long m_idBase;
    std::vector<cvtmi_opq_s *> m_hs;      // multi-device mode: one handle and one communicator per device
    std::vector<cvtmi_comm_s *> m_comms;
    long long m_devCap;
    std::vector<float> m_coarse, m_books;
    std::vector<float> m_R;               // dense rotation (SetRotation): replaces the permutation when present
    std::vector<int> m_reorder;
    int m_coarseK, m_pq_m, m_pq_k, m_pq_step, m_featDim, m_imgNum, m_maxIndexNum, m_imgCap;
};

// k smallest (score, index) ascending: get_sort_results of opq/src/common.h:25-37 (device top-k merge)
std::vector<std::pair<float, unsigned> > get_sort_results(const std::vector<float> &match_score, int results_per_query);
std::string get_base_name(const std::string path);
void get_vector_of_strings_from_file_lines(const std::string file_name, std::vector<std::string> &out);

template <typename T> void Delete2DArray(T **&f)
{
    if (f != NULL) {
        if (f[0] != NULL) { delete[] f[0]; f[0] = NULL; }
        delete[] f;
        f = NULL;
    }
}
template <typename T> void Init2DArray(T **&f, int row, int col)
{
    T *pf = new T[(size_t)row * col]();
    f = new T *[row];
    for (int i = 0; i < row; i++) f[i] = pf + (size_t)i * col;
}
#endif
