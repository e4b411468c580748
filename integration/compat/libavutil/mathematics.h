#pragma once
#include "common.h"
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
