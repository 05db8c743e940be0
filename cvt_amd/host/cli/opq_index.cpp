// opq_index -- the index-building main of the reference (opq/src/multi_frame_index_test.cpp:8-28) with real
// arguments instead of hard-coded paths:   opq_index <model> <feat_list.txt> <out_dir> [maxImageNum=1000]
#include <iostream>
#include "../IVFOPQ.h"
using namespace std;
int main(int argc, char *argv[])
{
    if (argc < 4) {
        cerr << "usage: opq_index <model> <feat_list.txt> <out_dir> [maxImageNum]" << endl;
        return 2;
    }
    string modelFile = argv[1], imgLists = argv[2], desDir = argv[3];
    int maxImageNum = argc > 4 ? atoi(argv[4]) : 1000;
    vector<string> featFiles;
    get_vector_of_strings_from_file_lines(imgLists, featFiles);
    cout << (int)featFiles.size() << endl;
    IVFOPQ index(maxImageNum);
    if (index.LoadModel(modelFile) != 1) return 1;
    index.IndexDatabase(featFiles);
    index.SaveIndex(desDir);
    return 0;
}
