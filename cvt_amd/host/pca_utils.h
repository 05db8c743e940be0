// pca_utils.h -- mirror of cvtk::PCAUtils (pca_train_project/pca_online/pca_utils.h:11-33; the same operation
// as PCAModel, pca_train_project/project/pca_dimension.h:7-58) above the C ABI.  Same methods; OpenCV is not a
// dependency: matrices are the plain fp32 row-major cvtk::Mat32f below, and loadModel reads the OpenCV
// FileStorage YAML the reference's models are stored in (`vectors`, `values`, `mean` as !!opencv-matrix nodes,
// e.g. pca_train_project/model/pca_1024_128_300w_googlenet.yml) with its own small reader.
// reduceDim runs on the MI355X (cvtmi_pca_project); there is no host projection.
#pragma once
#include <string>
#include <vector>

namespace cvtk {

struct Mat32f {  // fp32, row-major, contiguous: what the reference uses cv::Mat (CV_32FC1) for
    int rows = 0, cols = 0;
    std::vector<float> data;
    void create(int r, int c) { rows = r; cols = c; data.assign((size_t)r * c, 0.0f); }
    float &at(int i, int j) { return data[(size_t)i * cols + j]; }
    const float &at(int i, int j) const { return data[(size_t)i * cols + j]; }
    const float *row(int i) const { return data.data() + (size_t)i * cols; }
    bool empty() const { return data.empty(); }
};

// one `name: !!opencv-matrix` node (rows, cols, dt f|d, data [...]) out of FileStorage YAML text; throws
// std::runtime_error when the node is missing or malformed
Mat32f read_opencv_matrix(const std::string &yaml_text, const std::string &name);

class PCAUtils {
public:
    static PCAUtils *getInstance()
    {
        static PCAUtils inst;
        return &inst;
    }
    void loadModel(const std::string &filename);  // throws std::runtime_error on a missing / malformed file

    void reduceDim(const float *data, int num, int dim, Mat32f &reduceMat);  // NOLINT
    void reduceDim(const Mat32f &mat, Mat32f &reduceMat);                    // NOLINT
    Mat32f reduceDim(const Mat32f &mat);

    // the model (cv::PCA's public members in the reference)
    Mat32f eigenvectors, eigenvalues, mean;
};

}  // namespace cvtk
