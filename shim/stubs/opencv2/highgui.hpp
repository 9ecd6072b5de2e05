// mini-OpenCV: see opencv2/opencv.hpp
#pragma once
#include <opencv2/opencv.hpp>
