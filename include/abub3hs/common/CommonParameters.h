// number of post-trigger frames a bubble is tracked for (reference common/CommonParameters.h:5)
#ifndef ABUB3HS_COMMONPARAMETERS_H
#define ABUB3HS_COMMONPARAMETERS_H
#define NumFramesBubbleTrack 10
#endif
