// TEST INFRASTRUCTURE: declarations only — kdtree_tensor.hpp names the index type in an alias;
// model.cpp only calls PointsTensor::scales(), which oracle/ref_model_shim.cpp defines by brute force.
#pragma once
#include <cstddef>
namespace nanoflann {
template <class T, class DataSource, typename DistanceType = T, typename IndexType = size_t>
struct L2_Simple_Adaptor;
template <typename Distance, class DatasetAdaptor, int DIM = -1, typename IndexType = size_t>
class KDTreeSingleIndexAdaptor;
}  // namespace nanoflann
