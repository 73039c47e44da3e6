// PLY point colours of the GTSAM-side wrappers (gtsam/color.h:12-15: same enum, in namespace CG).  The g2o side's
// color.h of the reference has the same name but no namespace; both mirrors live in this one directory, hence the
// different file name.
#ifndef FGO_HOST_GT_COLOR_H
#define FGO_HOST_GT_COLOR_H
namespace CG {
typedef enum { RED = 0, GREEN, BLUE, PURPLE, WHITE, YELLOW, DARK } COLOR;
extern unsigned char g_color[][3];
}
#endif
