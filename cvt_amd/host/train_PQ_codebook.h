// TrainPQ -- host mirror of the reference's codebook trainer (opq/train_codebook/train_PQ_codebook.h:24-53)
// on top of the C ABI (cvtmi_opq_train).  Same constructor arguments, same public methods, same input
// (reorder file of D `long int`, raw fp32 feature file) and output (SaveCodebook model file that
// IVFOPQ::LoadModel reads) formats.  The k-means itself is the GPU Lloyd iteration documented in
// include/cvtmi.h: the reference's yael kmeans is not vendored, so centroids are not comparable bit for bit.
#ifndef CVTMI_HOST_TRAIN_PQ_CODEBOOK_H
#define CVTMI_HOST_TRAIN_PQ_CODEBOOK_H

#include <cstdint>
#include <string>
#include <vector>

class TrainPQ {
public:
    TrainPQ(std::string modelFile, int maxTrainFeatNum = 0, int featDim = 128, int coarseK = 8192, int pq_k = 256,
            int pq_m = 16);
    ~TrainPQ();
    void LoadFeatureSample(std::string srcDir);   // reads min(file rows, maxTrainFeatNum) rows, permuted by reorder
    void IFVPQ();                                 // CoarseQuan() + ProdQuan()
    void CoarseQuan();
    void ProdQuan();
    void SaveCodebook(std::string desDir);        // <desDir>/OPQ_db_<n>_dim_<D>_k_<coarseK>_PQ_m<M>_k<K>.fvecs
    void reorder(float *feat);

    // extra (not in the reference, which takes its "rotation" -- a permutation -- from the reorder file): learn a dense D x D rotation
    // from the loaded sample (cvtmi_opq_learn_rotation: `outer` rounds of per-sub-space k-means + orthogonal Procrustes) together with
    // the sub-codebooks.  Needs coarseK == 1 (the exhaustive configuration; the coarse centroid is the zero vector).  SaveCodebook then
    // writes the model as usual and the rotation beside it as raw fp32 [D][D]: <model>.R.f32 (IVFOPQ::LoadRotation).  1 ok / 0 failure.
    int LearnRotation(int outer);
    const std::vector<float> &rotation() const { return m_R; }
    // extras (not in the reference): what was produced, for callers that stay in process
    const std::string &modelPath() const { return m_desDir; }
    int featNum() const { return m_featNum; }
    int niter = 0;            // 0 = until convergence (the reference passes niter = 0 to yael)
    uint64_t seed = 1;        // the reference passes seed = 1 to every kmeans call

private:
    void train();
    std::string m_srcDir, m_desDir;
    int m_maxTrainFeatNum, m_featNum = 0, m_featDim, m_coarseK, m_pq_k, m_pq_m, m_pq_step;
    std::vector<float> m_feat;      // [featNum][featDim], permuted
    std::vector<float> m_coarse;    // [coarseK][featDim]
    std::vector<float> m_books;     // [pq_m][pq_k][pq_step]
    std::vector<float> m_R;         // [featDim][featDim], LearnRotation only
    std::vector<long int> reorder_; // as read from the reorder file
    bool m_trained = false;
};

#endif
