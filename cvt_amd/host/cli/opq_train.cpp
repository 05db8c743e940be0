// opq_train -- the reference's train_PQ main (opq/train_codebook/train_PQ.cpp:3-31), same 8 arguments:
//   opq_train <reorder file> <feature file> <output dir> <max samples> <coarseK> <featDim> <pq_m> <pq_k> [--learn-rotation=<outer>]
// --learn-rotation (not in the reference; coarseK must be 1): also learn a dense rotation from the sample (TrainPQ::LearnRotation) and
// write it beside the model as <model>.R.f32 -- opq_search --rotation reads it.
#include <cstdlib>
#include <iostream>
#include <string>

#include "../train_PQ_codebook.h"

int main(int argc, char *argv[])
{
    int learn = -1;
    if (argc == 10 && std::string(argv[9]).rfind("--learn-rotation=", 0) == 0) { learn = atoi(argv[9] + 17); --argc; }
    if (argc != 9) {
        std::cout << "Error in input parameters!\n";
        return -1;
    }
    const std::string modelPath = argv[1], srcDir = argv[2], desDir = argv[3];
    const int maxSampleNum = atoi(argv[4]), k = atoi(argv[5]), featDim = atoi(argv[6]), pq_m = atoi(argv[7]),
              pq_k = atoi(argv[8]);
    TrainPQ trainer(modelPath, maxSampleNum, featDim, k, pq_k, pq_m);
    trainer.LoadFeatureSample(srcDir);
    if (learn >= 0) {
        if (!trainer.LearnRotation(learn)) return 1;
    } else {
        trainer.IFVPQ();
    }
    trainer.SaveCodebook(desDir);
    std::cout << trainer.modelPath() << std::endl;
    return 0;
}
