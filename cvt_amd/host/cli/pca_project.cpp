// pca_project -- pca_train_project/pca_online/demo.cc:29-66 (and project/demo.cpp) as a tool:
//   pca_project <model.yml> <feats.txt> <out.txt>        project + L2-normalise every row of feats.txt
//   pca_project <model.yml> --info                        print the model's shape (no GPU involved)
// feats.txt: one row per line, "name,v0,v1,..." (the reference's feature dumps, e.g. project/data/1_test.txt);
// out.txt: "name v0 v1 ..." per row with 9 significant digits (what py/pca_compute.py:47-48 prints).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <vector>

#include "../pca_utils.h"

int main(int argc, char *argv[])
{
    if (argc < 3) {
        std::cout << "usage: pca_project <model.yml> <feats.txt> <out.txt> | pca_project <model.yml> --info\n";
        return -1;
    }
    try {
        cvtk::PCAUtils &model = *cvtk::PCAUtils::getInstance();
        model.loadModel(argv[1]);
        if (!strcmp(argv[2], "--info")) {
            double sv = 0, sm = 0;
            for (float v : model.eigenvectors.data) sv += v;
            for (float v : model.mean.data) sm += v;
            printf("vectors %d x %d sum %.9g; values %d x %d; mean %d x %d sum %.9g\n", model.eigenvectors.rows, model.eigenvectors.cols, sv,
                   model.eigenvalues.rows, model.eigenvalues.cols, model.mean.rows, model.mean.cols, sm);
            return 0;
        }
        if (argc < 4) { std::cout << "missing <out.txt>\n"; return -1; }
        std::ifstream fin(argv[2]);
        if (!fin) { std::cout << "cannot open " << argv[2] << "\n"; return 1; }
        const int dim = model.eigenvectors.cols;
        std::vector<std::string> names;
        std::vector<float> feats;
        std::string line;
        while (std::getline(fin, line)) {
            if (line.empty()) continue;
            const size_t comma = line.find(',');
            if (comma == std::string::npos) { std::cout << "bad line: " << line.substr(0, 40) << "\n"; return 1; }
            names.push_back(line.substr(0, comma));
            const char *c = line.c_str() + comma + 1;
            int got = 0;
            while (*c) {
                char *next = NULL;
                const float v = strtof(c, &next);  // std::stof in the reference (demo.cc:15)
                if (next == c) break;
                feats.push_back(v);
                ++got;
                c = (*next == ',') ? next + 1 : next;
            }
            if (got != dim) { std::cout << "row '" << names.back() << "' holds " << got << " values, the model takes " << dim << "\n"; return 1; }
        }
        cvtk::Mat32f out;
        model.reduceDim(feats.data(), (int)names.size(), dim, out);
        FILE *f = fopen(argv[3], "w");
        if (!f) { std::cout << "cannot write " << argv[3] << "\n"; return 1; }
        for (size_t i = 0; i < names.size(); ++i) {
            fprintf(f, "%s", names[i].c_str());
            for (int j = 0; j < out.cols; ++j) fprintf(f, " %.9g", (double)out.at((int)i, j));
            fprintf(f, "\n");
        }
        fclose(f);
        std::cout << names.size() << " rows projected " << dim << " -> " << out.cols << std::endl;
    } catch (const std::exception &e) {
        std::cout << "error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
