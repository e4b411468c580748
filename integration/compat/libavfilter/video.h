#pragma once
#include "avfilter.h"
