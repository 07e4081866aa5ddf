// stand-in for <opencv2/calib3d/calib3d.hpp>: see ../../mini_cv.h
#include "../../mini_cv.h"
