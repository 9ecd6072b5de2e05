// ORACLE — TEST INFRASTRUCTURE ONLY.  DBoW2's vocabulary descent as AirSLAM runs it, from the REFERENCE'S OWN sources compiled unchanged
// (oracle/Makefile): 3rdparty/DBoW2/include/DBoW2/TemplatedVocabulary.h (the template, instantiated here for FSuperpoint),
// 3rdparty/DBoW2/src/{BowVector,FeatureVector,ScoringObject}.cpp and src/bow/FSuperpoint.cc (FSuperpoint::distance).
// The vocabulary file voc/point_voc_L4.bin is absent upstream (.MISSING_LARGE_BLOBS): the tree is built in memory from flat arrays —
// the layout airfe_bow_load takes (children of a node contiguous, node 0 = root).
// What runs is TemplatedVocabulary::transform(feature, word_id, weight) (TemplatedVocabulary.h:1313-1352) per feature, called exactly as
// Database::FrameToBow does (src/bow/database.cc:57-89: `_voc->transform(features_eigen.block(3, i, 256, 1), id, w)`), then its w > 0 rule,
// BowVector::addWeight and the normalisation of :78-88.
#include <climits>
#include <cstdint>
#include <vector>

#include "3rdparty/DBoW2/include/DBoW2/TemplatedVocabulary.h"
#include "include/bow/FSuperpoint.h"

typedef DBoW2::TemplatedVocabulary<DBoW2::FSuperpoint::TDescriptor, DBoW2::FSuperpoint> SuperpointVocabulary;

extern "C" {
// node_desc [n][256], first_child / n_children / word_id [n] (word_id: of a leaf), weight [n].  feat [N][259] rows (= columns of the 259 x N matrix).
// out: word_of_features [N] (UINT_MAX where w <= 0: src/bow/database.cc:71-76), weight_of_features [N] (the w transform returned);
// bow_ids / bow_values (capacity N): the BowVector after FrameToBow's normalisation, ascending word id; returns its size.
int airslam_ref_bow_frame_to_bow(const float* node_desc, const int32_t* first_child, const int32_t* n_children, const int32_t* word_id, const double* weight,
                                 int n_nodes, int k, int L, const float* feat, int N, uint32_t* word_of_features, double* weight_of_features,
                                 uint32_t* bow_ids, double* bow_values) {
  SuperpointVocabulary voc(k, L, DBoW2::TF_IDF, DBoW2::L1_NORM);
  voc.m_nodes.resize((size_t)n_nodes);
  for (int i = 0; i < n_nodes; ++i) {
    auto& nd = voc.m_nodes[(size_t)i];
    nd.id = (DBoW2::NodeId)i;
    nd.weight = weight[i];
    nd.word_id = (DBoW2::WordId)(word_id[i] < 0 ? 0 : word_id[i]);
    for (int e = 0; e < 256; ++e) nd.descriptor(e, 0) = node_desc[(size_t)i * 256 + e];
    nd.children.clear();
    for (int c = 0; c < n_children[i]; ++c) {
      nd.children.push_back((DBoW2::NodeId)(first_child[i] + c));
      voc.m_nodes[(size_t)(first_child[i] + c)].parent = (DBoW2::NodeId)i;
    }
  }
  Eigen::Matrix<float, 259, Eigen::Dynamic> features_eigen;
  features_eigen.resize(259, N);
  for (int i = 0; i < N; ++i)
    for (int r = 0; r < 259; ++r) features_eigen(r, i) = feat[(size_t)i * 259 + r];
  // ---- src/bow/database.cc:57-89, statement for statement (the member function itself needs Frame / Database: not compiled)
  DBoW2::BowVector bow_vector;
  if (N == 0) return 0;
  DBoW2::LNorm norm;
  bool must = voc.m_scoring_object->mustNormalize(norm);
  for (int i = 0; i < N; i++) {
    DBoW2::WordId id;
    DBoW2::WordValue w;
    voc.transform(features_eigen.block(3, i, 256, 1), id, w);
    weight_of_features[i] = w;
    if (w > 0) {
      bow_vector.addWeight(id, w);
      word_of_features[i] = id;
    } else {
      word_of_features[i] = UINT_MAX;
    }
  }
  if (bow_vector.empty()) return 0;
  if (must) {
    bow_vector.normalize(norm);
  } else {
    const double nd = bow_vector.size();
    for (DBoW2::BowVector::iterator vit = bow_vector.begin(); vit != bow_vector.end(); vit++) vit->second /= nd;
  }
  int n = 0;
  for (const auto& kv : bow_vector) { bow_ids[n] = kv.first; bow_values[n] = kv.second; ++n; }
  return n;
}
}  // extern "C"
