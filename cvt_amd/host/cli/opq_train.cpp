// opq_train -- the reference's train_PQ main (opq/train_codebook/train_PQ.cpp:3-31), same 8 arguments:
//   opq_train <reorder file> <feature file> <output dir> <max samples> <coarseK> <featDim> <pq_m> <pq_k>
#include <cstdlib>
#include <iostream>
#include <string>

#include "../train_PQ_codebook.h"

int main(int argc, char *argv[])
{
    if (argc != 9) {
        std::cout << "Error in input parameters!\n";
        return -1;
    }
    const std::string modelPath = argv[1], srcDir = argv[2], desDir = argv[3];
    const int maxSampleNum = atoi(argv[4]), k = atoi(argv[5]), featDim = atoi(argv[6]), pq_m = atoi(argv[7]),
              pq_k = atoi(argv[8]);
    TrainPQ trainer(modelPath, maxSampleNum, featDim, k, pq_k, pq_m);
    trainer.LoadFeatureSample(srcDir);
    trainer.IFVPQ();
    trainer.SaveCodebook(desDir);
    std::cout << trainer.modelPath() << std::endl;
    return 0;
}
