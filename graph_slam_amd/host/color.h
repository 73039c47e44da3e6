// PLY point colours used by trajectoryPLY (mirrors the enum order of the reference's g2o/color.h:13-14, which
// drivers pass by name: BLUE / RED in g2o/test_g2o_graph.cpp:110,122).
#ifndef FGO_HOST_COLOR_H
#define FGO_HOST_COLOR_H
typedef enum { RED = 0, GREEN, BLUE, PURPLE, WHITE, YELLOW, DARK } COLOR;
extern unsigned char g_color[][3];
#endif
