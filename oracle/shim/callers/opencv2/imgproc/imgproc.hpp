// stand-in (declarations only): the OpenCV geometry calls of Frontend::doSetupAndInitialKeyframeDecision
#pragma once
#include <opencv2/core/core.hpp>
#include <vector>
namespace cv {
template <class P> void convexHull(const std::vector<P>& points, std::vector<P>& hull, bool clockwise = false, bool returnPoints = true);
template <class P> double contourArea(const std::vector<P>& contour, bool oriented = false);
template <class P, class Q> double pointPolygonTest(const std::vector<P>& contour, const Q& pt, bool measureDist);
}
