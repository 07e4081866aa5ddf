// stand-in for <opencv2/highgui/highgui.hpp>: see ../../mini_cv.h
#include "../../mini_cv.h"
