// opq_query -- the query main of the reference (opq/src/multi_frame_index_test.cpp:32-91):
//   opq_query <model> <index.fvecs> <result.txt> <query_feat> [more query files...] [--nearest 3] [--show 5]
#include <fstream>
#include <iostream>
#include <cstring>
#include "../IVFOPQ.h"
using namespace std;
int main(int argc, char *argv[])
{
    int num_nearest = 3, num_show = 5;
    vector<string> pos;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--nearest") && i + 1 < argc) num_nearest = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--show") && i + 1 < argc) num_show = atoi(argv[++i]);
        else pos.push_back(argv[i]);
    }
    if (pos.size() < 4) {
        cerr << "usage: opq_query <model> <index.fvecs> <result.txt> <query_feat>... [--nearest 3] [--show 5]" << endl;
        return 2;
    }
    IVFOPQ ivfpq_search;
    if (ivfpq_search.LoadModel(pos[0]) != 1) return 1;
    ivfpq_search.LoadIndex(pos[1]);
    ofstream fout(pos[2].c_str());
    vector<string> queryPaths(pos.begin() + 3, pos.end());
    for (int i = 0; i < (int)queryPaths.size(); i++) {
        vector<vector<float> > score;
        ivfpq_search.Query(queryPaths.at(i), score, num_nearest);
        int frame_num = (int)score.size();
        if (frame_num > 0) {
            int img_num = (int)score.at(0).size();
            vector<float> score_total(img_num, 0.0f);
            for (int j = 0; j < frame_num; j++)
                for (int k = 0; k < img_num; k++) score_total.at(k) += score.at(j).at(k);
            int show = min(num_show, img_num);
            vector<pair<float, unsigned> > result = get_sort_results(score_total, show);
            fout << queryPaths.at(i) << "  " << i << endl;
            for (int j = 0; j < show; j++) fout << get_base_name(ivfpq_search.m_imgLocation[result.at(j).second].ptr) << " ";
            fout << endl;
            for (int j = 0; j < show; j++) {
                fout << result.at(j).first << " ";
                std::cout << result.at(j).second << ": " << result.at(j).first << std::endl;
            }
            fout << "\n\n";
        }
        cout << "query ID: " << i << ", frame_num:" << frame_num << endl;
    }
    fout.close();
    return 0;
}
