// stand-in (declarations only): the OpenCV GUI calls of ThreadedKFVio::display
#pragma once
#include <opencv2/core/core.hpp>
#include <string>
namespace cv {
void namedWindow(const std::string& name, int flags = 1);
void imshow(const std::string& name, const Mat& image);
int waitKey(int delay = 0);
}
