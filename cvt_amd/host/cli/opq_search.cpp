// opq_search -- the north-star form of the OPQ query: exhaustive ADC top-k per query VECTOR over an index built from one
// raw feature file, on 1..N GPUs of one node.  Not a main of the reference (its query main scores videos,
// opq/src/multi_frame_index_test.cpp:32-91 = opq_query here); the multi-GPU shape is the reference tree's FLANN-MPI
// search (retrieval/vlindex/lib/FLANN/mpi/index.h:196-226): every rank indexes a contiguous block of rows, searches it,
// then ONE all-gather of the per-shard top-k + merge (inside libcvtmi: cvtmi_opq_search_sharded).
//
//   opq_search <model> <db_feat.bin> <query_feat.bin> <result.txt> [--k 100] [--gpus N] [--fork] [--transport rccl|shm] [--rotation R.f32]
//
// --rotation: a dense D x D rotation (raw fp32, e.g. opq_train --learn-rotation) instead of the model's permutation (IVFOPQ::LoadRotation).
// model: LoadModel format with coarseK == 1; feature files: raw fp32 [n][D] (IVFOPQ.cpp:451-457).
// --gpus N: ONE process drives the N GPUs (IVFOPQ::SetDevices: ncclCommInitAll + grouped all-gathers inside libcvtmi) -- the
// shape of the reference's own single-process mains.  --fork: one process per GPU instead (rank r -> device r %
// device_count, forked BEFORE any GPU work; rank 0 draws the RCCL id and shares it through a process-shared page).
// --transport shm (implies --fork) exchanges through host shared memory instead of RCCL (cvtmi_comm_create_custom): the
// stand-in for an MPI job, and how N ranks run on a box with ONE GPU (RCCL refuses two ranks on one device).
// A rank that fails before or inside a collective cannot leave the others waiting for ever: the library carries every
// rank's status through the all-gather (cvtmi.h), ranks waiting for the id give up when rank 0 reports failure, and the
// parent kills the surviving ranks as soon as one child exits with an error.  result.txt (rank 0): one line per query, "<qid> topK: id ... dists: d ..." like gt.txt
// (brute_force.cpp:106).
#include <pthread.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../../include/cvtmi.h"
#include "../IVFOPQ.h"
using namespace std;

static string g_rotation;   // --rotation: dense rotation file (IVFOPQ::LoadRotation) applied after every LoadModel

struct Shared {
    pthread_barrier_t bar;
    volatile int id_ready;
    char id[CVTMI_COMM_ID_BYTES];
    volatile int failed;
    size_t stage_bytes;
    // followed by the staging area of the shm transport
};
struct ShmCtx { Shared *sh; char *stage; int rank, world; };

// caller-supplied all-gather (cvtmi_allgather_fn): D2H of my slot, barrier, H2D of everybody's, barrier
static int shm_allgather(void *vctx, const void *send_dev, void *recv_dev, size_t bytes, void *stream)
{
    ShmCtx *c = (ShmCtx *)vctx;
    if (bytes * c->world > c->sh->stage_bytes) return 1;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 2;
    if (hipMemcpy(c->stage + (size_t)c->rank * bytes, send_dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 3;
    pthread_barrier_wait(&c->sh->bar);
    if (hipMemcpy(recv_dev, c->stage, bytes * c->world, hipMemcpyHostToDevice) != hipSuccess) return 4;
    pthread_barrier_wait(&c->sh->bar);
    return 0;
}

static long long file_rows(const string &path, int D)
{
    struct stat st;
    if (stat(path.c_str(), &st) != 0 || D <= 0) return -1;
    return (long long)(st.st_size / (sizeof(float) * D));
}

static int run_rank(int rank, int world, bool use_shm, Shared *sh, const string &model, const string &db, const string &qf,
                    const string &out, int k)
{
    // rank 0 owes the others an answer about the id on EVERY path out of this function
    struct IdGuard {
        Shared *sh; bool armed;
        ~IdGuard() { if (armed && !sh->id_ready) { sh->failed = 1; __sync_synchronize(); sh->id_ready = 1; } }
    } guard = { sh, rank == 0 };
    {   // inputs are checked before anybody enters a collective
        int D0 = 0;
        ifstream fm(model.c_str(), ios::binary);
        fm.read((char *)&D0, sizeof(int));
        if (!fm || D0 <= 0 || file_rows(db, D0) < 0 || file_rows(qf, D0) < 0) { fprintf(stderr, "rank %d: cannot read the model / feature files\n", rank); return 1; }
    }
    int ndev = 0;
    if (cvtmi_device_count(&ndev) != CVTMI_OK || ndev < 1) { fprintf(stderr, "rank %d: no GPU: %s\n", rank, cvtmi_last_error()); return 1; }
    if (cvtmi_set_device(rank % ndev) != CVTMI_OK) return 1;
    cvtmi_comm_t comm = NULL;
    ShmCtx ctx = { sh, (char *)(sh + 1), rank, world };
    if (world > 1 || !use_shm) {
        if (use_shm) {
            if (cvtmi_comm_create_custom(shm_allgather, &ctx, rank, world, &comm) != CVTMI_OK) { fprintf(stderr, "%s\n", cvtmi_last_error()); return 1; }
        } else {
            if (rank == 0) {
                if (cvtmi_comm_unique_id(sh->id) != CVTMI_OK) { fprintf(stderr, "%s\n", cvtmi_last_error()); sh->failed = 1; }
                __sync_synchronize();
                sh->id_ready = 1;
            }
            for (int waited_ms = 0; !sh->id_ready; ++waited_ms) {
                if (waited_ms > 120000) { fprintf(stderr, "rank %d: no RCCL id from rank 0 after 120 s\n", rank); return 1; }
                usleep(1000);
            }
            if (sh->failed) { fprintf(stderr, "rank %d: rank 0 could not provide the RCCL id\n", rank); return 1; }
            if (world == 1) cvtmi_set_tuning("comm_force_rccl", 1);  // --transport rccl was asked for: go through it even alone
            if (cvtmi_comm_create(sh->id, rank, world, &comm) != CVTMI_OK) { fprintf(stderr, "rank %d: %s\n", rank, cvtmi_last_error()); return 1; }
        }
    }
    IVFOPQ index;
    if (index.LoadModel(model) != 1) return 1;
    if (!g_rotation.empty() && index.LoadRotation(g_rotation) != 1) return 1;
    const int D = index.dim();
    const long long n = file_rows(db, D), nq = file_rows(qf, D);
    if (n < 0 || nq < 0) { fprintf(stderr, "cannot stat the feature files\n"); return 1; }
    int64_t a = 0, b = n;
    cvtmi_shard_range(n, rank, world, &a, &b);
    index.SetShard(comm, a);
    {   // this rank's block of rows, in chunks
        ifstream fin(db.c_str(), ios::binary);
        fin.seekg((std::streamoff)a * D * sizeof(float));
        const long long chunk = 1 << 18;
        vector<float> buf;
        for (long long r = a; r < b; r += chunk) {
            const long long m = min(chunk, (long long)b - r);
            buf.resize((size_t)m * D);
            fin.read((char *)buf.data(), sizeof(float) * buf.size());
            if (!fin || index.AddRows(buf.data(), (int)m) != 1) { fprintf(stderr, "rank %d: indexing failed\n", rank); return 1; }
        }
    }
    vector<float> q((size_t)nq * D);
    {
        ifstream fin(qf.c_str(), ios::binary);
        fin.read((char *)q.data(), sizeof(float) * q.size());
    }
    vector<float> dist((size_t)nq * k);
    vector<long long> ids((size_t)nq * k);
    if (nq > 0 && index.SearchTopK(q.data(), (int)nq, k, dist.data(), ids.data()) != 1) {
        fprintf(stderr, "rank %d: search failed: %s\n", rank, index.lastError().c_str());
        return 1;
    }
    if (rank == 0) {
        ofstream fout(out.c_str());
        char num[64];
        for (long long i = 0; i < nq; ++i) {
            fout << i << " topK: ";
            for (int j = 0; j < k; ++j) fout << ids[(size_t)i * k + j] << " ";
            fout << "dists: ";
            for (int j = 0; j < k; ++j) { snprintf(num, sizeof num, "%.9g ", dist[(size_t)i * k + j]); fout << num; }
            fout << "\n";
        }
        int64_t nc = 0, bytes = 0; int tr = 0;
        if (comm) cvtmi_comm_info(comm, NULL, NULL, &tr, &nc, &bytes);
        cout << "opq_search: " << n << " rows over " << world << " rank(s), rank 0 holds [" << a << ", " << b << "), " << nq
             << " queries, top-" << k << ", transport " << (tr == 1 ? "rccl" : tr == 2 ? "shm" : "none") << ", all-gathers " << nc
             << " x " << bytes << " B per rank" << endl;
    }
    if (comm) cvtmi_comm_destroy(comm);
    return 0;
}

// one process, `gpus` devices: IVFOPQ::SetDevices
static int run_single_process(int gpus, const string &model, const string &db, const string &qf, const string &out, int k)
{
    IVFOPQ index;
    if (index.LoadModel(model) != 1) return 1;
    if (!g_rotation.empty() && index.LoadRotation(g_rotation) != 1) return 1;
    const int D = index.dim();
    const long long n = file_rows(db, D), nq = file_rows(qf, D);
    if (n < 0 || nq < 0) { fprintf(stderr, "cannot stat the feature files\n"); return 1; }
    if (index.SetDevices(gpus, n) != 1) return 1;
    {
        ifstream fin(db.c_str(), ios::binary);
        const long long chunk = 1 << 18;
        vector<float> buf;
        for (long long r = 0; r < n; r += chunk) {
            const long long m = min(chunk, n - r);
            buf.resize((size_t)m * D);
            fin.read((char *)buf.data(), sizeof(float) * buf.size());
            if (!fin || index.AddRows(buf.data(), (int)m) != 1) { fprintf(stderr, "indexing failed\n"); return 1; }
        }
    }
    vector<float> q((size_t)nq * D);
    {
        ifstream fin(qf.c_str(), ios::binary);
        fin.read((char *)q.data(), sizeof(float) * q.size());
    }
    vector<float> dist((size_t)nq * k);
    vector<long long> ids((size_t)nq * k);
    if (nq > 0 && index.SearchTopK(q.data(), (int)nq, k, dist.data(), ids.data()) != 1) {
        fprintf(stderr, "search failed: %s\n", index.lastError().c_str());
        return 1;
    }
    ofstream fout(out.c_str());
    char num[64];
    for (long long i = 0; i < nq; ++i) {
        fout << i << " topK: ";
        for (int j = 0; j < k; ++j) fout << ids[(size_t)i * k + j] << " ";
        fout << "dists: ";
        for (int j = 0; j < k; ++j) { snprintf(num, sizeof num, "%.9g ", dist[(size_t)i * k + j]); fout << num; }
        fout << "\n";
    }
    cout << "opq_search: " << n << " rows over " << index.numDevices() << " device(s) of one process, " << nq << " queries, top-" << k << endl;
    return 0;
}

int main(int argc, char *argv[])
{
    int k = 100, gpus = 1;
    bool forked = false;
    string transport = "rccl";
    vector<string> pos;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--k") && i + 1 < argc) k = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--fork")) forked = true;
        else if (!strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--transport") && i + 1 < argc) transport = argv[++i];
        else if (!strcmp(argv[i], "--rotation") && i + 1 < argc) g_rotation = argv[++i];
        else pos.push_back(argv[i]);
    }
    if (pos.size() != 4 || k < 1 || k > 128 || gpus < 1 || (transport != "rccl" && transport != "shm")) {
        cerr << "usage: opq_search <model> <db_feat.bin> <query_feat.bin> <result.txt> [--k 100] [--gpus N] [--fork] [--transport rccl|shm] [--rotation R.f32]" << endl;
        return 2;
    }
    const bool use_shm = transport == "shm";
    if (!use_shm && !forked) return run_single_process(gpus, pos[0], pos[1], pos[2], pos[3], k);
    // staging area of the shm transport: world slots of one [nq][k] (f32, i64) result; D comes from the model header
    size_t stage = 0;
    if (use_shm) {
        int D = 0;
        ifstream fin(pos[0].c_str(), ios::binary);
        fin.read((char *)&D, sizeof(int));
        const long long nq = file_rows(pos[2], D);
        if (!fin || nq < 0) { cerr << "cannot read the model / query file" << endl; return 1; }
        size_t slot = 0;
        if (cvtmi_comm_slot_bytes(nq, k, &slot) != CVTMI_OK) { cerr << cvtmi_last_error() << endl; return 1; }
        stage = (size_t)gpus * slot;
    }
    // no GPU work before the fork: the children each own a fresh HIP runtime
    Shared *sh = (Shared *)mmap(NULL, sizeof(Shared) + stage, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (sh == MAP_FAILED) { perror("mmap"); return 1; }
    memset(sh, 0, sizeof(Shared));
    sh->stage_bytes = stage;
    pthread_barrierattr_t ba;
    pthread_barrierattr_init(&ba);
    pthread_barrierattr_setpshared(&ba, PTHREAD_PROCESS_SHARED);
    pthread_barrier_init(&sh->bar, &ba, gpus);
    if (gpus == 1) return run_rank(0, 1, use_shm, sh, pos[0], pos[1], pos[2], pos[3], k);
    vector<pid_t> kids;
    for (int r = 0; r < gpus; ++r) {
        pid_t p = fork();
        if (p < 0) { perror("fork"); return 1; }
        if (p == 0) _exit(run_rank(r, gpus, use_shm, sh, pos[0], pos[1], pos[2], pos[3], k));
        kids.push_back(p);
    }
    // a rank that dies leaves its peers in a barrier or a collective: the first failure ends them all
    int rc = 0;
    for (size_t left = kids.size(); left > 0; --left) {
        int st = 0;
        const pid_t p = waitpid(-1, &st, 0);
        if (p < 0) break;
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
            if (rc == 0)
                for (pid_t o : kids)
                    if (o != p) kill(o, SIGKILL);
            rc = 1;
        }
    }
    return rc;
}
