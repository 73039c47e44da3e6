// Stand-in for VRO's CCameraNode: ids + matchNodePair, served from the synthetic world.
#pragma once
#include "matching_result.h"

class CCameraNode {
 public:
  CCameraNode() : m_id(-1), m_seq_id(-1), m_frame(-1) {}
  virtual ~CCameraNode() {}
  int m_id;        // graph id, assigned by the graph wrapper
  int m_seq_id;    // sequence id
  int m_frame;     // synthetic frame index (set by CSparseFeatureVO::featureExtraction)
  // relative pose of `this` expressed in `older` (edge older -> this), as VRO's RANSAC would return it
  MatchingResult matchNodePair(CCameraNode *older);
};
