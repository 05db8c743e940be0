#include "train_PQ_codebook.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "../../include/cvtmi.h"

TrainPQ::TrainPQ(std::string modelFile, int maxTrainFeatNum, int featDim, int coarseK, int pq_k, int pq_m)
    : m_maxTrainFeatNum(maxTrainFeatNum), m_featDim(featDim), m_coarseK(coarseK), m_pq_k(pq_k), m_pq_m(pq_m)
{
    m_pq_step = m_featDim / m_pq_m;
    printf("feature dimension: %d\n", featDim);
    // reorder file: featDim values of `long int` (train_PQ_codebook.cpp:15-18); a missing file leaves zeros there
    reorder_.assign((size_t)m_featDim, 0);
    std::ifstream fin(modelFile, std::ios::binary);
    if (fin) fin.read(reinterpret_cast<char *>(reorder_.data()), sizeof(long int) * (size_t)m_featDim);
    for (int i = 0; i < m_featDim; i++) std::cout << reorder_[(size_t)i] << " ";
    std::cout << std::endl;
}

TrainPQ::~TrainPQ() {}

void TrainPQ::reorder(float *feat)
{
    if (feat == NULL) return;
    std::vector<float> tmp((size_t)m_featDim);
    for (int i = 0; i < m_featDim; i++) tmp[(size_t)i] = feat[reorder_[(size_t)i]];
    std::memcpy(feat, tmp.data(), sizeof(float) * (size_t)m_featDim);
}

void TrainPQ::LoadFeatureSample(std::string srcDir)
{
    m_srcDir = srcDir;
    FILE *f = fopen(m_srcDir.c_str(), "rb");
    if (f == NULL) {
        std::cout << "Fail to open the source file " << srcDir << std::endl;
        return;
    }
    std::cout << "Load training features, please wait...\n";
    fseek(f, 0, SEEK_END);
    const long long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    const long long rows = bytes / (long long)(sizeof(float) * (size_t)m_featDim);
    m_featNum = (int)std::min<long long>(rows, m_maxTrainFeatNum);  // train_PQ_codebook.cpp:63
    m_feat.assign((size_t)m_featNum * m_featDim, 0.0f);
    std::vector<float> row((size_t)m_featDim);
    for (int r = 0; r < m_featNum; r++) {
        if (fread(row.data(), sizeof(float), (size_t)m_featDim, f) != (size_t)m_featDim) { m_featNum = r; break; }
        float *dst = &m_feat[(size_t)r * m_featDim];
        for (int n = 0; n < m_featDim; n++) dst[n] = row[(size_t)reorder_[(size_t)n]];  // :80
    }
    fclose(f);
    m_feat.resize((size_t)m_featNum * m_featDim);
    std::cout << m_featNum << " training features loaded!\n";
}

void TrainPQ::train()
{
    if (m_trained) return;
    m_coarse.assign((size_t)m_coarseK * m_featDim, 0.0f);
    m_books.assign((size_t)m_pq_m * m_pq_k * m_pq_step, 0.0f);
    if (m_featNum <= 0) return;
    const int rc = cvtmi_opq_train(m_feat.data(), m_featNum, m_featDim, m_coarseK, m_pq_m, m_pq_k, niter, seed,
                                   m_coarse.data(), m_books.data());
    if (rc != CVTMI_OK) {
        std::cout << "training failed: " << cvtmi_last_error() << std::endl;
        return;
    }
    m_trained = true;
}

int TrainPQ::LearnRotation(int outer)
{
    if (m_coarseK != 1) { std::cout << "LearnRotation: needs coarseK == 1" << std::endl; return 0; }
    if (m_featNum <= 0) return 0;
    m_coarse.assign((size_t)m_featDim, 0.0f);
    m_books.assign((size_t)m_pq_m * m_pq_k * m_pq_step, 0.0f);
    m_R.assign((size_t)m_featDim * m_featDim, 0.0f);
    const int rc = cvtmi_opq_learn_rotation(m_feat.data(), m_featNum, m_featDim, m_pq_m, m_pq_k, outer, niter, seed, m_R.data(), m_books.data());
    if (rc != CVTMI_OK) {
        std::cout << "rotation learning failed: " << cvtmi_last_error() << std::endl;
        m_R.clear();
        return 0;
    }
    m_trained = true;
    return 1;
}

// The reference trains the coarse quantiser and the sub-quantisers in two calls; here both come out of one
// device-side pass (the residuals never leave HBM), run by whichever is called first.
void TrainPQ::CoarseQuan()
{
    train();
    std::cout << "finish coarse quantization." << std::endl;
}

void TrainPQ::ProdQuan()
{
    std::cout << "product quantization......." << std::endl;
    train();
}

void TrainPQ::IFVPQ()
{
    CoarseQuan();
    ProdQuan();
}

void TrainPQ::SaveCodebook(std::string desDir)
{
    std::ostringstream name;
    name << desDir << "/OPQ_db_" << m_featNum << "_dim_" << m_featDim << "_k_" << m_coarseK << "_PQ_m" << m_pq_m << "_k"
         << m_pq_k << ".fvecs";
    m_desDir = name.str();
    std::ofstream out(m_desDir.c_str(), std::ios::binary);
    out.write(reinterpret_cast<const char *>(&m_featDim), sizeof(int));
    out.write(reinterpret_cast<const char *>(&m_coarseK), sizeof(int));
    out.write(reinterpret_cast<const char *>(&m_pq_m), sizeof(int));
    out.write(reinterpret_cast<const char *>(&m_pq_k), sizeof(int));
    out.write(reinterpret_cast<const char *>(m_coarse.data()), sizeof(float) * m_coarse.size());
    out.write(reinterpret_cast<const char *>(m_books.data()), sizeof(float) * m_books.size());
    // The reference writes sizeof(int) * featDim BYTES of its `long int` array (:287), i.e. the first half of
    // it, and IVFOPQ::LoadModel reads those bytes as int32[featDim] (IVFOPQ.cpp:94-95).  Kept byte for byte:
    // the file stays interchangeable with the reference's tools, quirk included.
    out.write(reinterpret_cast<const char *>(reorder_.data()), sizeof(int) * (size_t)m_featDim);
    out.close();
    if (!m_R.empty()) {   // LearnRotation: the dense rotation beside the model (the model format has no slot for it)
        // The sample was permuted by reorder_ when it was loaded (LoadFeatureSample, :80), so R and the codebooks were learned for
        // y = R (P x) with (P x)[n] = x[reorder_[n]].  IVFOPQ::LoadRotation applies the file's matrix to the RAW vector and drops the
        // permutation (a handle has a rotation or a permutation), so the file carries the product R P:  R'[i][reorder_[n]] = R[i][n].
        std::vector<float> folded(m_R.size(), 0.0f);
        std::vector<char> seen((size_t)m_featDim, 0);
        for (int n = 0; n < m_featDim; n++) {   // (LoadFeatureSample indexed the rows with these values: they are in range if we got here)
            const long int j = reorder_[(size_t)n];
            if (j < 0 || j >= m_featDim || seen[(size_t)j]) {
                std::cout << "SaveCodebook: the reorder file is not a permutation of 0.." << m_featDim - 1 << ": rotation not written" << std::endl;
                return;
            }
            seen[(size_t)j] = 1;
        }
        for (int i = 0; i < m_featDim; i++)
            for (int n = 0; n < m_featDim; n++)
                folded[(size_t)i * m_featDim + (size_t)reorder_[(size_t)n]] = m_R[(size_t)i * m_featDim + n];
        std::ofstream rout((m_desDir + ".R.f32").c_str(), std::ios::binary);
        rout.write(reinterpret_cast<const char *>(folded.data()), sizeof(float) * folded.size());
    }
}
