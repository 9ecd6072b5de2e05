// ORACLE — TEST INFRASTRUCTURE ONLY.  Stand-in for the reference's include/line_processor.h (which pulls in g2o, OpenCV
// ximgproc and camera.h): declarations of the functions of src/line_processor.cc:1-180 — the part oracle/Makefile extracts
// and compiles — with the reference's signatures (include/line_processor.h:20-33).
#ifndef LINE_PROCESSOR_H_
#define LINE_PROCESSOR_H_
#include <Eigen/Dense>
#include <map>
#include <opencv2/opencv.hpp>
#include <string>
#include <vector>

#include "utils.h"

float PointLineDistance(Eigen::Vector4f line, Eigen::Vector2f point);
double CVPointLineDistance3D(const std::vector<cv::Point3f> points, const cv::Vec6f& line, std::vector<float>& dist);
void EigenPointLineDistance3D(const std::vector<Eigen::Vector3d>& points, const Vector6d& line, std::vector<double>& dist);
float AngleDiff(float& angle1, float& angle2);
void AssignPointsToLines(std::vector<Eigen::Vector4d>& lines, Eigen::Matrix<float, 259, Eigen::Dynamic>& points,
                         std::vector<std::map<int, double>>& relation);
void MatchLines(const std::vector<std::map<int, double>>& points_on_line0, const std::vector<std::map<int, double>>& points_on_line1,
                const std::vector<cv::DMatch>& point_matches, size_t point_num0, size_t point_num1, std::vector<int>& line_matches);
#endif  // LINE_PROCESSOR_H_
