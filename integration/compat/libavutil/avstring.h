#pragma once
#include "common.h"
