// stand-in (declarations only) for OpenGV's opengv/types.hpp: the vector / matrix typedefs of the published API
#pragma once
#include <vector>
#include <Eigen/Core>
#include <Eigen/StdVector>
namespace opengv {
typedef Eigen::Vector3d bearingVector_t;
typedef std::vector<bearingVector_t, Eigen::aligned_allocator<bearingVector_t> > bearingVectors_t;
typedef Eigen::Vector3d translation_t;
typedef std::vector<translation_t, Eigen::aligned_allocator<translation_t> > translations_t;
typedef Eigen::Matrix3d rotation_t;
typedef std::vector<rotation_t, Eigen::aligned_allocator<rotation_t> > rotations_t;
typedef Eigen::Matrix<double, 3, 4> transformation_t;
typedef std::vector<transformation_t, Eigen::aligned_allocator<transformation_t> > transformations_t;
typedef Eigen::Vector3d cayley_t;
typedef Eigen::Vector4d quaternion_t;
typedef Eigen::Matrix3d essential_t;
typedef std::vector<essential_t, Eigen::aligned_allocator<essential_t> > essentials_t;
typedef Eigen::Vector3d point_t;
typedef std::vector<point_t, Eigen::aligned_allocator<point_t> > points_t;
}
