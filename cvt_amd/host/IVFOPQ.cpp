// IVFOPQ.cpp -- host side of the reference's IVFOPQ class above the C ABI (see IVFOPQ.h).
#include "IVFOPQ.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "../../include/cvtmi.h"

using namespace std;

void IVFOPQ::init()
{
    m_h = NULL; m_imgLocation = NULL; m_comm = NULL; m_idBase = 0; m_devCap = 0;
    m_coarseK = m_pq_m = m_pq_k = m_pq_step = m_featDim = 0;
    m_imgNum = 0; m_imgCap = 0;
}
IVFOPQ::IVFOPQ(int maxIndexNum) : m_maxIndexNum(maxIndexNum) { init(); }
IVFOPQ::IVFOPQ() : m_maxIndexNum(1 << 30) { init(); }
IVFOPQ::~IVFOPQ()
{
    for (size_t d = 1; d < m_hs.size(); ++d) cvtmi_opq_destroy(m_hs[d]);   // (m_hs[0] == m_h)
    for (size_t d = 0; d < m_comms.size(); ++d) cvtmi_comm_destroy(m_comms[d]);
    if (m_h) cvtmi_opq_destroy(m_h);
    delete[] m_imgLocation;
}
std::string IVFOPQ::lastError() const { return cvtmi_last_error(); }
long long IVFOPQ::numEntries() const
{
    int64_t total = 0, n = 0;
    if (m_hs.size() > 1) {
        for (size_t d = 0; d < m_hs.size(); ++d) { n = 0; cvtmi_opq_ntotal(m_hs[d], &n); total += n; }
        return total;
    }
    if (m_h) cvtmi_opq_ntotal(m_h, &n);
    return n;
}

int IVFOPQ::SetDevices(int ndev, long long expected_rows)
{
    if (ndev < 1 || expected_rows < 0 || !ensureHandle() || m_hs.size() > 1 || m_comm) { printf("SetDevices: bad state or arguments\n"); return 0; }
    if (numEntries() != 0) { printf("SetDevices: the index already holds entries\n"); return 0; }
    if (ndev == 1) return 1;
    int avail = 0;
    if (cvtmi_device_count(&avail) != CVTMI_OK || avail < ndev) { printf("SetDevices: %d devices asked for, %d present\n", ndev, avail); return 0; }
    m_devCap = std::max<long long>(1, (expected_rows + ndev - 1) / ndev);
    // device 0 must be the one m_h lives on: re-create the handles in device order
    cvtmi_opq_destroy(m_h); m_h = NULL;
    std::vector<cvtmi_opq_s *> hs(ndev, (cvtmi_opq_s *)NULL);
    bool ok = true;
    for (int d = 0; d < ndev && ok; ++d) {
        ok = cvtmi_set_device(d) == CVTMI_OK &&
             cvtmi_opq_create(m_featDim, m_coarseK, m_pq_m, m_pq_k, m_coarse.data(), m_books.data(), m_R.empty() ? NULL : m_R.data(),
                              (!m_R.empty() || m_reorder.empty()) ? NULL : m_reorder.data(), &hs[d]) == CVTMI_OK &&
             cvtmi_opq_set_id_base(hs[d], (long long)d * m_devCap) == CVTMI_OK;
    }
    std::vector<cvtmi_comm_s *> cs(ndev, (cvtmi_comm_s *)NULL);
    ok = ok && cvtmi_comm_create_all(ndev, NULL, cs.data()) == CVTMI_OK;
    cvtmi_set_device(0);
    if (!ok) {
        printf("SetDevices failed: %s\n", cvtmi_last_error());
        for (int d = 0; d < ndev; ++d) { if (hs[d]) cvtmi_opq_destroy(hs[d]); if (cs[d]) cvtmi_comm_destroy(cs[d]); }
        ensureHandle();
        return 0;
    }
    m_hs = hs; m_comms = cs; m_h = hs[0];
    return 1;
}

int IVFOPQ::SetRotation(const float *R)
{
    if (!R || m_featDim <= 0) { printf("SetRotation: load a model first\n"); return 0; }
    if (m_hs.size() > 1 || (m_h && numEntries() != 0)) { printf("SetRotation: the index already holds entries\n"); return 0; }
    m_R.assign(R, R + (size_t)m_featDim * m_featDim);
    if (m_h) { cvtmi_opq_destroy(m_h); m_h = NULL; }   // (re-created with the rotation on first use)
    return ensureHandle() ? 1 : 0;
}

int IVFOPQ::LoadRotation(std::string rotationFile)
{
    ifstream fin(rotationFile.c_str(), ios::binary | ios::ate);
    if (!fin.is_open() || m_featDim <= 0) { printf("Can not open the rotation file!\n"); return 0; }
    const size_t want = sizeof(float) * (size_t)m_featDim * m_featDim;
    if ((size_t)fin.tellg() != want) { printf("rotation file: %zu bytes expected\n", want); return 0; }
    fin.seekg(0);
    std::vector<float> R((size_t)m_featDim * m_featDim);
    fin.read((char *)R.data(), (std::streamsize)want);
    return SetRotation(R.data());
}

bool IVFOPQ::ensureHandle()
{
    if (m_h) return true;
    if (m_featDim <= 0) return false;
    int rc = cvtmi_opq_create(m_featDim, m_coarseK, m_pq_m, m_pq_k, m_coarse.data(), m_books.data(), m_R.empty() ? NULL : m_R.data(),
                              (!m_R.empty() || m_reorder.empty()) ? NULL : m_reorder.data(), &m_h);
    if (rc != CVTMI_OK) {
        printf("cvtmi_opq_create failed: %s\n", cvtmi_last_error());
        m_h = NULL;
        return false;
    }
    if (m_idBase) cvtmi_opq_set_id_base(m_h, m_idBase);
    return true;
}

// model file: int32 D, coarseK, M, K; fp32 coarse[coarseK][D]; fp32 books[M][K][D/M]; int32 reorder[D]
int IVFOPQ::LoadModel(std::string modelFile)
{
    printf("load training file....\n");
    ifstream fin(modelFile.c_str(), ios::binary);
    if (!fin.is_open()) {
        printf("Can not open the model file!\n");
        return 0;
    }
    fin.read((char *)&m_featDim, sizeof(int));
    fin.read((char *)&m_coarseK, sizeof(int));
    fin.read((char *)&m_pq_m, sizeof(int));
    fin.read((char *)&m_pq_k, sizeof(int));
    if (!fin || m_featDim <= 0 || m_coarseK <= 0 || m_pq_m <= 0 || m_pq_k <= 0 || m_featDim % m_pq_m) {
        printf("Bad model header!\n");
        return 0;
    }
    m_pq_step = m_featDim / m_pq_m;
    printf("feature dimension: %d, coarse codebook size: %d, subspace dimension: %d, number of subspace codebook: %d, "
           "finetune codebook size: %d\n", m_featDim, m_coarseK, m_pq_step, m_pq_m, m_pq_k);
    m_coarse.resize((size_t)m_coarseK * m_featDim);
    fin.read((char *)m_coarse.data(), sizeof(float) * m_coarse.size());
    m_books.resize((size_t)m_featDim * m_pq_k);
    fin.read((char *)m_books.data(), sizeof(float) * m_books.size());
    m_reorder.resize(m_featDim);
    fin.read((char *)m_reorder.data(), sizeof(int) * m_featDim);
    if (!fin) {
        printf("Model file truncated!\n");
        return 0;
    }
    fin.close();
    if (m_h) { cvtmi_opq_destroy(m_h); m_h = NULL; }
    return ensureHandle() ? 1 : 0;
}

// IVFOPQ::Add: coarse argmin + residual PQ argmin per row on the GPU, entries appended with
// videoId = m_imgNum (IVFOPQ.cpp:105-170)
void IVFOPQ::Add(float **m_ppFeat, const int m_frameNum)
{
    if (m_frameNum <= 0 || !ensureHandle()) return;
    std::vector<int32_t> lists(m_frameNum), vids(m_frameNum, m_imgNum);
    std::vector<uint8_t> codes((size_t)m_frameNum * m_pq_m);
    // Init2DArray rows are one contiguous block (common.h:62-72); tolerate scattered rows as well
    const float *x = m_ppFeat[0];
    std::vector<float> tmp;
    bool contiguous = true;
    for (int i = 1; i < m_frameNum && contiguous; ++i) contiguous = m_ppFeat[i] == m_ppFeat[0] + (size_t)i * m_featDim;
    if (!contiguous) {
        tmp.resize((size_t)m_frameNum * m_featDim);
        for (int i = 0; i < m_frameNum; ++i) memcpy(&tmp[(size_t)i * m_featDim], m_ppFeat[i], sizeof(float) * m_featDim);
        x = tmp.data();
    }
    if (cvtmi_opq_encode(m_h, x, m_frameNum, lists.data(), codes.data()) != CVTMI_OK ||
        cvtmi_opq_add_codes(m_h, codes.data(), m_coarseK > 1 ? lists.data() : NULL, vids.data(), m_frameNum) != CVTMI_OK)
        printf("Add failed: %s\n", cvtmi_last_error());
}

void IVFOPQ::IndexDatabase(vector<string> featFiles)
{
    int num = min(int(featFiles.size()), m_maxIndexNum);
    delete[] m_imgLocation;
    m_imgLocation = new ImgNameStruct[num > 0 ? num : 1];
    m_imgCap = num;
    for (int i = 0; i < num; i++) {
        std::cout << featFiles.at(i) << std::endl;
        float **m_ppFeat = NULL;
        int m_frameNum = 0;
        std::string srcFile = featFiles.at(i);
        LoadSingleFeatFile(srcFile, m_ppFeat, m_frameNum);
        if (m_frameNum > 0) {
            Add(m_ppFeat, m_frameNum);
            memset(m_imgLocation[m_imgNum].ptr, 0, max_path * sizeof(char));
            strncpy(m_imgLocation[m_imgNum].ptr, srcFile.c_str(), max_path - 1);
            m_imgNum++;
        }
        Delete2DArray(m_ppFeat);
    }
}

void IVFOPQ::queryImpl(const std::string &featFile, vector<vector<float> > &matchScore, int nk)
{
    float **m_ppFeat = NULL;
    int m_frameNum = 0;
    LoadSingleFeatFile(featFile, m_ppFeat, m_frameNum);
    if (m_frameNum == 0) return;
    matchScore.resize(m_frameNum);
    std::vector<float> ms((size_t)m_frameNum * std::max(m_imgNum, 1));
    int rc = CVTMI_OK;
    if (m_imgNum > 0) rc = cvtmi_opq_query_video(m_h, m_ppFeat[0], m_frameNum, /*rotate=*/0, nk, m_imgNum, ms.data());
    if (rc != CVTMI_OK) printf("Query failed: %s\n", cvtmi_last_error());
    for (int f = 0; f < m_frameNum; ++f)
        matchScore.at(f).assign(ms.begin() + (size_t)f * m_imgNum, ms.begin() + (size_t)(f + 1) * m_imgNum);
    Delete2DArray(m_ppFeat);
}
void IVFOPQ::Query(string featFile, vector<vector<float> > &matchScore, int nk) { queryImpl(featFile, matchScore, nk); }
void IVFOPQ::QueryThrehold(string featFile, vector<vector<float> > &matchScore, int nk) { queryImpl(featFile, matchScore, nk); }

int IVFOPQ::SearchTopK(const float *q, int nq, int k, float *dist, long long *ids)
{
    if (!ensureHandle()) return 0;
    static_assert(sizeof(long long) == sizeof(int64_t), "id width");
    if (m_hs.size() > 1)
        return cvtmi_opq_search_sharded_all(m_hs.data(), m_comms.data(), (int)m_hs.size(), q, nq, /*rotate=*/1, k, dist, (int64_t *)ids) == CVTMI_OK ? 1 : 0;
    if (m_comm) return cvtmi_opq_search_sharded(m_h, m_comm, q, nq, /*rotate=*/1, k, dist, (int64_t *)ids) == CVTMI_OK ? 1 : 0;
    return cvtmi_opq_search(m_h, q, nq, /*rotate=*/1, k, dist, (int64_t *)ids) == CVTMI_OK ? 1 : 0;
}

void IVFOPQ::SetShard(cvtmi_comm_s *comm, long long id_base)
{
    m_comm = comm; m_idBase = id_base;
    if (ensureHandle()) cvtmi_opq_set_id_base(m_h, id_base);
}

int IVFOPQ::AddRows(const float *raw, int n)
{
    if (n <= 0) return 1;
    if (!ensureHandle()) return 0;
    if (m_hs.size() > 1) {   // fill the devices in order: device d owns ids [d cap, (d + 1) cap), the last one whatever is left
        long long done = 0;
        while (done < n) {
            int d = 0;
            long long held = 0;
            for (; d < (int)m_hs.size(); ++d) {
                int64_t have = 0;
                cvtmi_opq_ntotal(m_hs[d], &have);
                held = have;
                if (d + 1 == (int)m_hs.size() || have < m_devCap) break;
            }
            const long long room = d + 1 == (int)m_hs.size() ? (long long)n - done : std::min<long long>(m_devCap - held, (long long)n - done);
            cvtmi_opq_s *h = m_hs[d];
            std::vector<uint8_t> codes((size_t)room * m_pq_m);
            std::vector<int32_t> lists((size_t)room);
            std::vector<float> rot((size_t)room * m_featDim);
            const float *src = raw + (size_t)done * m_featDim;
            if (cvtmi_set_device(d) != CVTMI_OK || cvtmi_opq_rotate(h, src, room, rot.data()) != CVTMI_OK ||
                cvtmi_opq_encode(h, rot.data(), room, lists.data(), codes.data()) != CVTMI_OK ||
                cvtmi_opq_add_codes(h, codes.data(), m_coarseK > 1 ? lists.data() : NULL, NULL, room) != CVTMI_OK) {
                printf("AddRows (device %d) failed: %s\n", d, cvtmi_last_error());
                cvtmi_set_device(0);
                return 0;
            }
            done += room;
        }
        cvtmi_set_device(0);
        return 1;
    }
    std::vector<float> rot((size_t)n * m_featDim);
    std::vector<int32_t> lists(n);
    std::vector<uint8_t> codes((size_t)n * m_pq_m);
    if (cvtmi_opq_rotate(m_h, raw, n, rot.data()) != CVTMI_OK || cvtmi_opq_encode(m_h, rot.data(), n, lists.data(), codes.data()) != CVTMI_OK ||
        cvtmi_opq_add_codes(m_h, codes.data(), m_coarseK > 1 ? lists.data() : NULL, NULL, n) != CVTMI_OK) {
        printf("AddRows failed: %s\n", cvtmi_last_error());
        return 0;
    }
    return 1;
}

// raw fp32 [n][D] file, n = bytes / (4 D); every row goes through the model's rotation (:441-462)
void IVFOPQ::LoadSingleFeatFile(string srcFile, float **&m_ppFeat, int &m_frameNum)
{
    ifstream fin(srcFile.c_str(), ios::binary);
    if (!fin.is_open()) {
        cout << "Error open the feat file: " << srcFile << endl;
        return;
    }
    fin.seekg(0, ios::end);
    long long file_size = (long long)fin.tellg();
    fin.seekg(0, ios::beg);
    m_frameNum = m_featDim > 0 ? (int)(file_size / (sizeof(float) * m_featDim)) : 0;
    if (m_frameNum <= 0) { m_frameNum = 0; return; }
    std::vector<float> raw((size_t)m_frameNum * m_featDim);
    fin.read((char *)raw.data(), sizeof(float) * raw.size());
    fin.close();
    Init2DArray(m_ppFeat, m_frameNum, m_featDim);
    if (!ensureHandle() || cvtmi_opq_rotate(m_h, raw.data(), m_frameNum, m_ppFeat[0]) != CVTMI_OK) {
        printf("rotation failed: %s\n", cvtmi_last_error());
        Delete2DArray(m_ppFeat);
        m_frameNum = 0;
    }
}

static string index_file_name(const string &dir, int imgNum, int D, int K, int m, int k)
{
    stringstream ss;
    ss << dir << "/OPQ_Index_db_" << imgNum << "_dim_" << D << "_k_" << K << "_PQ_m" << m << "_k" << k << ".fvecs";
    return ss.str();
}

// int32 D, coarseK, M, K, imgNum; coarse; books; per list: int32 n, n x {int32 videoId, uint8 code[M]};
// imgNum x char[260]   (IVFOPQ.cpp:544-580).  log.txt is written like the reference does.
void IVFOPQ::SaveIndex(string desDir)
{
    cout << "save index file..." << endl;
    if (!ensureHandle()) return;
    string path = index_file_name(desDir, m_imgNum, m_featDim, m_coarseK, m_pq_m, m_pq_k);
    ofstream outFile(path.c_str(), ios::binary);
    outFile.write((char *)&m_featDim, sizeof(int));
    outFile.write((char *)&m_coarseK, sizeof(int));
    outFile.write((char *)&m_pq_m, sizeof(int));
    outFile.write((char *)&m_pq_k, sizeof(int));
    outFile.write((char *)&m_imgNum, sizeof(int));
    outFile.write((char *)m_coarse.data(), sizeof(float) * m_coarse.size());
    outFile.write((char *)m_books.data(), sizeof(float) * m_books.size());
    int64_t n = 0;
    cvtmi_opq_ntotal(m_h, &n);
    std::vector<int64_t> off((size_t)m_coarseK + 1);
    std::vector<int32_t> vid((size_t)n);
    std::vector<uint8_t> codes((size_t)n * m_pq_m);
    if (cvtmi_opq_get_entries(m_h, off.data(), vid.data(), codes.data()) != CVTMI_OK)
        printf("SaveIndex failed: %s\n", cvtmi_last_error());
    ofstream fout("log.txt");
    for (int i = 0; i < m_coarseK; i++) {
        int cnt = (int)(off[i + 1] - off[i]);
        if (cnt > 0) {
            fout << "coarse center: " << i << ", " << cnt << std::endl;
            std::cout << "coarse center: " << i << ", " << cnt << std::endl;
        }
        outFile.write((char *)&cnt, sizeof(int));
        for (int64_t j = off[i]; j < off[i + 1]; j++) {
            outFile.write((char *)&vid[j], sizeof(int));
            outFile.write((char *)&codes[(size_t)j * m_pq_m], m_pq_m);
        }
    }
    fout.close();
    for (int i = 0; i < m_imgNum; i++) outFile.write(m_imgLocation[i].ptr, sizeof(char) * max_path);
    outFile.close();
}

void IVFOPQ::LoadIndex(string srcFile)
{
    cout << "load index..." << endl;
    ifstream fin(srcFile.c_str(), ios::binary);
    if (!fin.is_open()) {
        cout << "Can not open the index file." << endl;
        exit(0);
    }
    fin.read((char *)&m_featDim, sizeof(int));
    fin.read((char *)&m_coarseK, sizeof(int));
    fin.read((char *)&m_pq_m, sizeof(int));
    fin.read((char *)&m_pq_k, sizeof(int));
    fin.read((char *)&m_imgNum, sizeof(int));
    if (!fin || m_featDim <= 0 || m_pq_m <= 0 || m_featDim % m_pq_m) {
        cout << "Bad index header." << endl;
        exit(0);
    }
    m_pq_step = m_featDim / m_pq_m;
    m_coarse.resize((size_t)m_coarseK * m_featDim);
    fin.read((char *)m_coarse.data(), sizeof(float) * m_coarse.size());
    m_books.resize((size_t)m_featDim * m_pq_k);
    fin.read((char *)m_books.data(), sizeof(float) * m_books.size());
    // the rotation is not part of the index file (the reference loads the model first, multi_frame_index_test.cpp:46-47)
    if ((int)m_reorder.size() != m_featDim) {
        m_reorder.resize(m_featDim);
        for (int i = 0; i < m_featDim; ++i) m_reorder[i] = i;
    }
    if (m_h) { cvtmi_opq_destroy(m_h); m_h = NULL; }
    if (!ensureHandle()) exit(0);
    std::vector<int32_t> lists, vids;
    std::vector<uint8_t> codes;
    for (int i = 0; i < m_coarseK; i++) {
        int cnt = 0;
        fin.read((char *)&cnt, sizeof(int));
        for (int j = 0; j < cnt; ++j) {
            int v = 0;
            fin.read((char *)&v, sizeof(int));
            size_t o = codes.size();
            codes.resize(o + m_pq_m);
            fin.read((char *)&codes[o], m_pq_m);
            lists.push_back(i); vids.push_back(v);
        }
    }
    if (!lists.empty() &&
        cvtmi_opq_add_codes(m_h, codes.data(), m_coarseK > 1 ? lists.data() : NULL, vids.data(), (int64_t)lists.size()) != CVTMI_OK)
        printf("LoadIndex failed: %s\n", cvtmi_last_error());
    delete[] m_imgLocation;
    m_imgLocation = new ImgNameStruct[m_imgNum > 0 ? m_imgNum : 1];
    m_imgCap = m_imgNum;
    for (int i = 0; i < m_imgNum; i++) fin.read(m_imgLocation[i].ptr, sizeof(char) * max_path);
    fin.close();
}

// ---- opq/src/common.h helpers ----
std::vector<std::pair<float, unsigned> > get_sort_results(const std::vector<float> &match_score, int results_per_query)
{
    // the k smallest (score, index) pairs, selected on the device
    const int n = (int)match_score.size();
    std::vector<std::pair<float, unsigned> > t(results_per_query > 0 ? results_per_query : 0);
    if (n == 0 || results_per_query <= 0) return t;
    // (the device selection takes k <= CVTMI_K_MAX = 2048; entries beyond stay value-initialised, as the reference's partial_sort_copy
    //  leaves those beyond match_score.size())
    const int k = results_per_query > CVTMI_K_MAX ? CVTMI_K_MAX : results_per_query;
    std::vector<float> od(k);
    std::vector<int64_t> oi(k);
    if (cvtmi_topk_select(match_score.data(), 1, n, k, od.data(), oi.data()) != CVTMI_OK) {
        printf("get_sort_results failed: %s\n", cvtmi_last_error());
        return t;
    }
    for (int i = 0; i < k; ++i) t[i] = std::make_pair(od[i], (unsigned)oi[i]);
    return t;
}

std::string get_base_name(const std::string path)
{
    size_t pos = path.find_last_of("/\\");
    std::string s1 = path.substr(pos + 1);
    pos = s1.find_last_of('.');
    return s1.substr(0, pos);
}

void get_vector_of_strings_from_file_lines(const std::string file_name, std::vector<std::string> &out)
{
    std::ifstream in_file(file_name.c_str());
    std::string line;
    out.clear();
    while (std::getline(in_file, line))
        if (!line.empty()) out.push_back(line);
}
