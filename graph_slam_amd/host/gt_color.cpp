#include "gt_color.h"
namespace CG {
// rows follow the COLOR enum
unsigned char g_color[][3] = {
    {255, 0, 0}, {0, 255, 0}, {0, 0, 255}, {255, 0, 255}, {255, 255, 255}, {255, 255, 0}, {0, 0, 0},
};
}
