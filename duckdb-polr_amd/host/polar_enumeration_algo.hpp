// duckdb-polr_amd/host/polar_enumeration_algo.hpp -- host mirror of the join-order enumerators
// (src/include/duckdb/parallel/polar_enumeration_algo.hpp:22-131, src/parallel/polar_enumeration_algo.cpp).
// Setup only (once per pipeline, host side); the "bank of alternative probe orders" the multiplexer
// routes over.  Path 0 is always the optimizer's original order.
#pragma once

#include <map>
#include <random>
#include <set>
#include <unordered_map>

#include "physical_hash_join.hpp"

namespace duckdb_polr {

// vocabulary of this header
using JoinOrder = vector<idx_t>;                             // a permutation (or a prefix of one) of join indices
using JoinList = vector<PhysicalHashJoin *>;                 // the multiplexed joins, original order
using DependencyMap = std::unordered_map<idx_t, JoinOrder>;  // join -> joins whose build columns it is keyed by

// ---- how a min-card / uncertain / random enumerator picks among the joins that may come next ------------------
class CandidateSelector {
public:
	virtual ~CandidateSelector() = default;
	virtual idx_t SelectNextCandidate(const JoinOrder &candidates,
	                                  const JoinList &joins_p) = 0;
};

class RandomCandidateSelector : public CandidateSelector {
public:
	idx_t SelectNextCandidate(const JoinOrder &candidates, const JoinList &joins) override;
};

class MinCardinalitySelector : public CandidateSelector {
public:
	idx_t SelectNextCandidate(const JoinOrder &candidates, const JoinList &joins) override;
};

// UncertainCardinalitySelector (polar_enumeration_algo.cpp:32-77) walks the build side's plan tree;
// the host mirror has no plan trees, so each join carries the level the walk would return
// (PhysicalHashJoin::uncertainty_level, 1 + #filters/joins below the build side).
class UncertainCardinalitySelector : public CandidateSelector {
public:
	idx_t SelectNextCandidate(const JoinOrder &candidates, const JoinList &joins) override;
	std::unordered_map<idx_t, idx_t> uncertainties;
};

// ---- the enumerators (JoinEnumerator values of join_enumerator.hpp:15-25) --------------------------------------
// base class = "only the original order"; max_join_orders caps the bank
class JoinEnumerationAlgo {
public:
	virtual ~JoinEnumerationAlgo() = default;
	virtual void GenerateJoinOrders(const vector<idx_t> &hash_join_idxs,
	                                DependencyMap &dependencies,
	                                const JoinList &joins, vector<JoinOrder> &join_orders);
	bool CanJoin(vector<idx_t> &r, idx_t s, DependencyMap &dependencies);
	bool CanJoin(vector<idx_t> &r, vector<idx_t> &s, DependencyMap &dependencies); // ANY of s may follow r (:137-145)
	static unique_ptr<JoinEnumerationAlgo> CreateEnumerationAlgo(ClientContext &context);
	idx_t max_join_orders = 24;
};

// SelSampleEnumeration (polar_enumeration_algo.hpp:85-100, polar_enumeration_algo.cpp:323-556) -- the reference's
// DEFAULT enumerator (client_config.hpp:90): max_join_orders rounds of DPsize over the joins, each round with freshly
// SAMPLED selectivities for the relations that carry a predicate, so that every round may crown a different order; the
// bank = the original order + the distinct winners, in lexicographic order.  Randomness: std::mt19937(1337) through
// std::uniform_real_distribution<double> -- the same libstdc++ classes here, hence the same stream.  Nodes are
// identified by their position (0 = the pipeline's source, 1 + j = join j): the reference keys its maps by node
// POINTERS into one contiguous vector, which order the same way.
class SelSampleEnumeration : public JoinEnumerationAlgo {
public:
	using NodeSet = std::set<idx_t>;
	using NodeSeq = vector<idx_t>;
	void GenerateJoinOrders(const vector<idx_t> &hash_join_idxs, DependencyMap &dependencies, const JoinList &joins,
	                        vector<JoinOrder> &join_orders) override;
	NodeSeq DpSize(const vector<JoinOrderNodeInfo> &nodes, DependencyMap &dependencies);
	double CalculateCost(const NodeSeq &join_order);

private:
	const vector<JoinOrderNodeInfo> *nodes = nullptr;
	std::map<NodeSeq, double> cost_map;
	std::map<NodeSet, double> card_map;
	std::map<NodeSet, NodeSeq> best_plans;
	std::mt19937 rng = std::mt19937(1337);
	std::uniform_real_distribution<double> dist;
	const vector<double> SEL_STEPS = {0.0001, 0.001, 0.01, 0.1, 0.2, 0.4, 0.8};
};

class DFSEnumeration : public JoinEnumerationAlgo {
public:
	explicit DFSEnumeration(unique_ptr<CandidateSelector> selector_p) : selector(std::move(selector_p)) {
	}
	void GenerateJoinOrders(const vector<idx_t> &hash_join_idxs, DependencyMap &dependencies,
	                        const JoinList &joins, vector<JoinOrder> &join_orders) override;
	void GeneratePathsRecursive(const JoinList &joins,
	                            DependencyMap &join_prerequisites,
	                            vector<vector<idx_t>> &result, vector<idx_t> join_seq, vector<idx_t> joins_left);
	const unique_ptr<CandidateSelector> selector;
};

class BFSEnumeration : public JoinEnumerationAlgo {
public:
	explicit BFSEnumeration(unique_ptr<CandidateSelector> selector_p) : selector(std::move(selector_p)) {
	}
	void GenerateJoinOrders(const vector<idx_t> &hash_join_idxs, DependencyMap &dependencies,
	                        const JoinList &joins, vector<JoinOrder> &join_orders) override;
	vector<idx_t> FindJoinCandidates(idx_t join_count, vector<idx_t> &predecessors,
	                                 DependencyMap &dependencies);
	const unique_ptr<CandidateSelector> selector;
};

class EachLastOnceEnumeration : public JoinEnumerationAlgo {
public:
	void GenerateJoinOrders(const vector<idx_t> &hash_join_idxs, DependencyMap &dependencies,
	                        const JoinList &joins, vector<JoinOrder> &join_orders) override;
};

class EachFirstOnceEnumeration : public JoinEnumerationAlgo {
public:
	void GenerateJoinOrders(const vector<idx_t> &hash_join_idxs, DependencyMap &dependencies,
	                        const JoinList &joins, vector<JoinOrder> &join_orders) override;
};

} // namespace duckdb_polr
