#include "polar_enumeration_algo.hpp"

#include <algorithm>
#include <limits>
#include <numeric>
#include <queue>

namespace duckdb_polr {

// Candidate selection (reference behaviour: polar_enumeration_algo.cpp:13-77).  Ties go to the candidate that
// comes first in `candidates`, as a strict "<" scan does.
idx_t RandomCandidateSelector::SelectNextCandidate(const JoinOrder &candidates, const JoinList &) {
	// libc rand(), like the reference: not reproducible across libcs
	return candidates[rand() % candidates.size()];
}

idx_t MinCardinalitySelector::SelectNextCandidate(const JoinOrder &candidates, const JoinList &joins) {
	if (candidates.empty()) {
		return 0;
	}
	return *std::min_element(candidates.begin(), candidates.end(), [&](idx_t a, idx_t b) {
		return joins[a]->estimated_cardinality < joins[b]->estimated_cardinality;
	});
}

idx_t UncertainCardinalitySelector::SelectNextCandidate(const JoinOrder &candidates, const JoinList &joins) {
	// a join's score is computed once and remembered: uncertainty level x estimated cardinality
	auto score = [&](idx_t j) {
		auto it = uncertainties.find(j);
		if (it == uncertainties.end()) {
			it = uncertainties.emplace(j, joins[j]->uncertainty_level * joins[j]->estimated_cardinality).first;
		}
		return it->second;
	};
	if (candidates.empty()) {
		return 0;
	}
	idx_t best = candidates.front();
	for (const idx_t j : candidates) {
		if (score(j) < score(best)) {
			best = j;
		}
	}
	return best;
}

// The enumerator the session asks for (reference behaviour: polar_enumeration_algo.cpp:79-124)
unique_ptr<JoinEnumerationAlgo> JoinEnumerationAlgo::CreateEnumerationAlgo(ClientContext &context) {
	if (context.config.join_enumerator == JoinEnumerator::SAMPLE) {
		unique_ptr<JoinEnumerationAlgo> sample(new SelSampleEnumeration());
		sample->max_join_orders = context.config.max_join_orders;
		return sample;
	}
	enum class Walk { DEPTH, BREADTH, LAST_ONCE, FIRST_ONCE };
	enum class Pick { NONE, RANDOM, MIN_CARD, UNCERTAIN };
	Walk walk;
	Pick pick = Pick::NONE;
	switch (context.config.join_enumerator) {
	case JoinEnumerator::DFS_RANDOM:      walk = Walk::DEPTH;   pick = Pick::RANDOM;    break;
	case JoinEnumerator::DFS_MIN_CARD:    walk = Walk::DEPTH;   pick = Pick::MIN_CARD;  break;
	case JoinEnumerator::DFS_UNCERTAIN:   walk = Walk::DEPTH;   pick = Pick::UNCERTAIN; break;
	case JoinEnumerator::BFS_RANDOM:      walk = Walk::BREADTH; pick = Pick::RANDOM;    break;
	case JoinEnumerator::BFS_MIN_CARD:    walk = Walk::BREADTH; pick = Pick::MIN_CARD;  break;
	case JoinEnumerator::BFS_UNCERTAIN:   walk = Walk::BREADTH; pick = Pick::UNCERTAIN; break;
	case JoinEnumerator::EACH_LAST_ONCE:  walk = Walk::LAST_ONCE;  break;
	case JoinEnumerator::EACH_FIRST_ONCE: walk = Walk::FIRST_ONCE; break;
	default:
		throw InternalException("unknown join enumerator");
	}
	unique_ptr<CandidateSelector> selector;
	if (pick == Pick::RANDOM) {
		selector.reset(new RandomCandidateSelector());
	} else if (pick == Pick::MIN_CARD) {
		selector.reset(new MinCardinalitySelector());
	} else if (pick == Pick::UNCERTAIN) {
		selector.reset(new UncertainCardinalitySelector());
	}
	unique_ptr<JoinEnumerationAlgo> algo;
	switch (walk) {
	case Walk::DEPTH:
		algo.reset(new DFSEnumeration(std::move(selector)));
		break;
	case Walk::BREADTH:
		algo.reset(new BFSEnumeration(std::move(selector)));
		break;
	case Walk::LAST_ONCE:
		algo.reset(new EachLastOnceEnumeration());
		break;
	default:
		algo.reset(new EachFirstOnceEnumeration());
		break;
	}
	algo->max_join_orders = context.config.max_join_orders;
	return algo;
}

// :126-135
bool JoinEnumerationAlgo::CanJoin(vector<idx_t> &r, idx_t s, DependencyMap &dependencies) {
	auto &prereq = dependencies[s];
	for (const auto required_relation : prereq) {
		if (std::find(r.begin(), r.end(), required_relation) == r.end()) {
			return false;
		}
	}
	return true;
}

// :137-145
bool JoinEnumerationAlgo::CanJoin(vector<idx_t> &r, vector<idx_t> &s, DependencyMap &dependencies) {
	for (auto &si : s) {
		if (CanJoin(r, si, dependencies)) {
			return true;
		}
	}
	return false;
}

// :137-150: just the default join order
void JoinEnumerationAlgo::GenerateJoinOrders(const vector<idx_t> &hash_join_idxs,
                                             DependencyMap &dependencies,
                                             const JoinList &joins,
                                             vector<JoinOrder> &join_orders) {
	std::vector<idx_t> default_path(hash_join_idxs.size());
	std::iota(default_path.begin(), default_path.end(), 0);
	join_orders.reserve(max_join_orders);
	join_orders.push_back(default_path);
}

// :152-189
void DFSEnumeration::GeneratePathsRecursive(const JoinList &joins,
                                            DependencyMap &join_prerequisites,
                                            vector<vector<idx_t>> &result, vector<idx_t> join_seq,
                                            vector<idx_t> joins_left) {
	if (result.size() >= max_join_orders) {
		return;
	}
	vector<idx_t> candidates;
	for (auto join_idx : joins_left) {
		if (CanJoin(join_seq, join_idx, join_prerequisites)) {
			candidates.push_back(join_idx);
		}
	}
	idx_t num_relations = candidates.size();
	for (idx_t i = 0; i < num_relations; i++) {
		const idx_t join_idx = selector->SelectNextCandidate(candidates, joins);
		candidates.erase(std::find(candidates.begin(), candidates.end(), join_idx));
		vector<idx_t> join_seq_new(join_seq);
		join_seq_new.push_back(join_idx);
		if (joins_left.size() == 1) {
			result.push_back(join_seq_new);
		} else {
			vector<idx_t> joins_left_new(joins_left);
			joins_left_new.erase(std::find(joins_left_new.begin(), joins_left_new.end(), join_idx));
			GeneratePathsRecursive(joins, join_prerequisites, result, std::move(join_seq_new),
			                       std::move(joins_left_new));
		}
	}
}

// the "original join order first" fix-up shared by DFS and BFS (:573-608, :717-747)
static void MoveOriginalOrderFirst(vector<JoinOrder> &join_orders, idx_t k, idx_t max_join_orders) {
	bool contains_original = false;
	idx_t original_idx = 0;
	for (idx_t i = 0; i < join_orders.size(); i++) {
		bool is_original = true;
		for (idx_t j = 0; j < join_orders[i].size(); j++) {
			if (join_orders[i][j] != j) {
				is_original = false;
				break;
			}
		}
		if (is_original) {
			contains_original = true;
			original_idx = i;
			break;
		}
	}
	if (!contains_original) {
		vector<idx_t> original(k);
		std::iota(original.begin(), original.end(), 0);
		join_orders.insert(join_orders.begin(), original);
		if (join_orders.size() > max_join_orders) {
			join_orders.erase(join_orders.end() - 1);
		}
	} else if (original_idx != 0) {
		auto original = join_orders[original_idx];
		join_orders.erase(join_orders.begin() + original_idx);
		join_orders.insert(join_orders.begin(), original);
	}
}

// :558-608
void DFSEnumeration::GenerateJoinOrders(const vector<idx_t> &hash_join_idxs,
                                        DependencyMap &dependencies,
                                        const JoinList &joins, vector<JoinOrder> &join_orders) {
	vector<idx_t> joins_left(hash_join_idxs.size());
	std::iota(joins_left.begin(), joins_left.end(), 0);
	join_orders.reserve(max_join_orders + 1);
	GeneratePathsRecursive(joins, dependencies, join_orders, vector<idx_t>(), joins_left);
	MoveOriginalOrderFirst(join_orders, hash_join_idxs.size(), max_join_orders);
}

// Append `sequence` to `orders` if every join in it comes after the joins it depends on.
static void AddIfValid(JoinEnumerationAlgo &algo, const JoinOrder &sequence, DependencyMap &dependencies,
                       vector<JoinOrder> &orders) {
	JoinOrder placed;
	for (const idx_t join : sequence) {
		if (!algo.CanJoin(placed, join, dependencies)) {
			return;
		}
		placed.push_back(join);
	}
	orders.push_back(placed);
}

// original order, then every order that moves ONE join to the end (the last join stays: that is the original)
// (reference behaviour: polar_enumeration_algo.cpp:610-638)
void EachLastOnceEnumeration::GenerateJoinOrders(const vector<idx_t> &hash_join_idxs, DependencyMap &dependencies,
                                                 const JoinList &joins, vector<JoinOrder> &join_orders) {
	JoinEnumerationAlgo::GenerateJoinOrders(hash_join_idxs, dependencies, joins, join_orders);
	const JoinOrder original = join_orders.front();
	for (idx_t moved = 0; moved + 1 < original.size(); moved++) {
		JoinOrder sequence;
		for (idx_t pos = 0; pos < original.size(); pos++) {
			if (pos != moved) {
				sequence.push_back(original[pos]);
			}
		}
		sequence.push_back(original[moved]);
		AddIfValid(*this, sequence, dependencies, join_orders);
	}
}

// original order, then every order that moves ONE join to the front (:640-667)
void EachFirstOnceEnumeration::GenerateJoinOrders(const vector<idx_t> &hash_join_idxs, DependencyMap &dependencies,
                                                  const JoinList &joins, vector<JoinOrder> &join_orders) {
	JoinEnumerationAlgo::GenerateJoinOrders(hash_join_idxs, dependencies, joins, join_orders);
	const JoinOrder original = join_orders.front();
	for (idx_t moved = 1; moved < original.size(); moved++) {
		JoinOrder sequence {original[moved]};
		for (idx_t pos = 0; pos < original.size(); pos++) {
			if (pos != moved) {
				sequence.push_back(original[pos]);
			}
		}
		AddIfValid(*this, sequence, dependencies, join_orders);
	}
}

// :669-685
struct JoinCandidateEntry {
	idx_t level;
	idx_t candidate_idx;
	idx_t step;
	vector<idx_t> predecessors;
	idx_t candidate;
	friend bool operator<(JoinCandidateEntry const &left, JoinCandidateEntry const &right) {
		if (left.level == right.level) {
			if (left.candidate_idx == right.candidate_idx) {
				return left.step > right.step;
			}
			return left.candidate_idx > right.candidate_idx;
		}
		return left.level > right.level;
	}
};

vector<idx_t> BFSEnumeration::FindJoinCandidates(idx_t join_count, vector<idx_t> &predecessors,
                                                 DependencyMap &dependencies) {
	vector<bool> found_relation(join_count, false);
	for (auto predecessor : predecessors) {
		found_relation[predecessor] = true;
	}
	vector<idx_t> result;
	for (idx_t i = 0; i < found_relation.size(); i++) {
		if (!found_relation[i] && CanJoin(predecessors, i, dependencies)) {
			result.push_back(i);
		}
	}
	return result;
}

// :687-747
void BFSEnumeration::GenerateJoinOrders(const vector<idx_t> &hash_join_idxs,
                                        DependencyMap &dependencies,
                                        const JoinList &joins, vector<JoinOrder> &join_orders) {
	std::priority_queue<JoinCandidateEntry> queue;
	vector<idx_t> empty_predecessors;
	vector<idx_t> first_level = FindJoinCandidates(hash_join_idxs.size(), empty_predecessors, dependencies);
	idx_t step = 0;
	idx_t num_initial = std::min<idx_t>(4, first_level.size());
	for (idx_t i = 0; i < num_initial; i++) {
		idx_t next = selector->SelectNextCandidate(first_level, joins);
		queue.push(JoinCandidateEntry {0, i, step, empty_predecessors, next});
		first_level.erase(std::find(first_level.begin(), first_level.end(), next));
		step++;
	}
	join_orders.reserve(max_join_orders + 1);
	while (join_orders.size() <= max_join_orders && !queue.empty()) {
		auto entry = queue.top();
		auto &predecessors = entry.predecessors;
		queue.pop();
		predecessors.push_back(entry.candidate);
		auto candidates = FindJoinCandidates(hash_join_idxs.size(), predecessors, dependencies);
		if (predecessors.size() == hash_join_idxs.size() - 1 && candidates.size() == 1) {
			predecessors.push_back(candidates.front());
			join_orders.push_back(predecessors);
		} else {
			idx_t num = (idx_t)std::max(1, 4 - (int)predecessors.size());
			num = std::min<idx_t>(num, candidates.size());
			for (idx_t i = 0; i < num; i++) {
				idx_t candidate = selector->SelectNextCandidate(candidates, joins);
				candidates.erase(std::find(candidates.begin(), candidates.end(), candidate));
				queue.push(JoinCandidateEntry {predecessors.size(), i, step, predecessors, candidate});
				step++;
			}
		}
	}
	MoveOriginalOrderFirst(join_orders, hash_join_idxs.size(), max_join_orders);
}

// ---- SelSampleEnumeration ------------------------------------------------------------------------------------
// GenerateQuantifierSets (:289-321): the r-subsets of {0..n-1} in lexicographic order
static void QuantifierSets(idx_t n, idx_t r, idx_t first, vector<idx_t> &cur, vector<vector<idx_t>> &out) {
	if (cur.size() == r) {
		out.push_back(cur);
		return;
	}
	for (idx_t i = first; i < n; i++) {
		cur.push_back(i);
		QuantifierSets(n, r, i + 1, cur, out);
		cur.pop_back();
	}
}

// :332-390
SelSampleEnumeration::NodeSeq SelSampleEnumeration::DpSize(const vector<JoinOrderNodeInfo> &initial_join_order,
                                                           DependencyMap &dependencies) {
	vector<idx_t> empty;
	const idx_t n = initial_join_order.size() - 1; // the joins
	NodeSet join_nodes;
	for (idx_t i = 1; i <= n; i++) {
		join_nodes.insert(i);
		if (CanJoin(empty, i - 1, dependencies)) {
			best_plans[NodeSet {i}] = NodeSeq {0, i};
		}
	}
	for (idx_t s = 1; s < n; s++) {
		vector<vector<idx_t>> qsets;
		vector<idx_t> cur;
		QuantifierSets(n, s, 0, cur, qsets);
		for (auto &p_s1 : qsets) {
			for (idx_t p_s2 = 0; p_s2 < n; p_s2++) {
				if (std::find(p_s1.begin(), p_s1.end(), p_s2) != p_s1.end()) {
					continue; // !Disjoint
				}
				if (!CanJoin(empty, p_s1, dependencies) || !CanJoin(p_s1, p_s2, dependencies)) {
					continue;
				}
				NodeSet new_set;
				for (auto idx : p_s1) {
					new_set.insert(idx + 1);
				}
				auto found = best_plans.find(new_set);
				if (found == best_plans.cend()) {
					continue;
				}
				NodeSeq new_plan = found->second;
				new_set.insert(p_s2 + 1);
				new_plan.push_back(p_s2 + 1);
				auto best_plan = best_plans.find(new_set);
				if (best_plan != best_plans.cend()) {
					// (which plan is costed first decides which one draws its selectivities first: part of the stream)
					const bool calc_new_plan_first = std::round(dist(rng)) != 0;
					double c_new, c_best;
					if (calc_new_plan_first) {
						c_new = CalculateCost(new_plan);
						c_best = CalculateCost(best_plan->second);
					} else {
						c_best = CalculateCost(best_plan->second);
						c_new = CalculateCost(new_plan);
					}
					if (c_new < c_best) {
						best_plans[new_set] = new_plan;
					}
				} else {
					best_plans[new_set] = new_plan;
				}
			}
		}
	}
	return best_plans[join_nodes];
}

// :404-483.  join_order.front() is always the source.  Kept as the reference has it, including the cast that binds
// before the multiplication in the last branch (`SEL_STEPS[(idx_t) rand * SEL_STEPS.size()]` is SEL_STEPS[0], :464).
double SelSampleEnumeration::CalculateCost(const NodeSeq &join_order) {
	auto entry = cost_map.find(join_order);
	if (entry != cost_map.cend()) {
		return entry->second;
	}
	const vector<JoinOrderNodeInfo> &N = *nodes;
	if (join_order.size() == 1) {
		const JoinOrderNodeInfo &node = N[join_order.front()];
		double card = (double)node.base_table_card;
		if (node.predicate) {
			auto rand = dist(rng);
			auto sel = SEL_STEPS[(idx_t)(rand * SEL_STEPS.size())] + rand * SEL_STEPS[0];
			card *= sel;
		}
		card_map[NodeSet(join_order.cbegin(), join_order.cend())] = card;
		cost_map[join_order] = 0;
	} else {
		NodeSet lhs(join_order.cbegin(), join_order.cend() - 1);
		NodeSeq lhs_ordered(join_order.begin(), join_order.cend() - 1);
		NodeSet rhs {join_order.back()};
		NodeSet new_set = lhs;
		new_set.insert(join_order.back());
		if (card_map.find(lhs) == card_map.cend()) {
			CalculateCost(lhs_ordered);
		}
		if (card_map.find(rhs) == card_map.cend()) {
			CalculateCost(NodeSeq {join_order.back()});
		}
		double card = card_map[lhs];
		// GetJoinsWithPredicate (:392-402): the relations of the new set that carry a predicate, plus the source
		NodeSet lhs_predicates_only;
		for (auto node : new_set) {
			if (N[node].predicate) {
				lhs_predicates_only.insert(node);
			}
		}
		lhs_predicates_only.insert(join_order.front());
		if (card_map.find(new_set) != card_map.cend()) {
			card = card_map[new_set];
		} else if (card_map.find(lhs_predicates_only) != card_map.cend()) {
			card = card_map[lhs_predicates_only];
		} else if (N[join_order.back()].unique) {
			// the largest cardinality already fixed for a superset bounds this one from below
			double min_card = 0;
			for (auto &card_entry : card_map) {
				if (card_entry.first.size() > new_set.size()) {
					bool is_superset = true;
					for (auto node : new_set) {
						if (card_entry.first.find(node) == card_entry.first.cend()) {
							is_superset = false;
						}
					}
					if (!is_superset) {
						continue;
					}
					min_card = card_entry.second > min_card ? card_entry.second : min_card;
				}
			}
			if (N[join_order.back()].predicate) {
				auto rand = dist(rng);
				auto sel = SEL_STEPS[(idx_t)(rand * SEL_STEPS.size())] + rand * SEL_STEPS[0];
				card = min_card + sel * (card - min_card);
			}
		} else {
			auto rand = dist(rng);
			auto sel = SEL_STEPS[(idx_t)rand * SEL_STEPS.size()] + rand * SEL_STEPS[0];
			card *= card_map[rhs] * sel;
		}
		card_map[new_set] = card;
		cost_map[join_order] = cost_map[lhs_ordered] + card;
	}
	return cost_map[join_order];
}

static idx_t Factorial(idx_t i) {
	return i <= 1 ? 1 : i * Factorial(i - 1);
}

// :492-556
void SelSampleEnumeration::GenerateJoinOrders(const vector<idx_t> &hash_join_idxs, DependencyMap &dependencies,
                                              const JoinList &joins, vector<JoinOrder> &join_orders) {
	const idx_t SAMPLE_COUNT = max_join_orders;
	// CreateJoinOrderNodes (:248-287): the source, then the build side of every join
	vector<JoinOrderNodeInfo> node_infos;
	node_infos.push_back(joins.front()->probe_source_info);
	for (auto *join : joins) {
		node_infos.push_back(join->build_side_info);
	}
	for (auto &ni : node_infos) {
		if (ni.nested) {
			throw NotImplementedException("SelSampleEnumeration over a nested join tree");
		}
	}
	nodes = &node_infos;
	idx_t rhs_relations_with_predicate = 0;
	for (idx_t i = 1; i < node_infos.size(); i++) {
		if (node_infos[i].predicate || !node_infos[i].unique) {
			rhs_relations_with_predicate++;
		}
	}
	const idx_t max_unique_join_orders = Factorial(rhs_relations_with_predicate);
	std::set<vector<idx_t>> unique_join_orders;
	vector<idx_t> inital_join_order(joins.size());
	std::iota(inital_join_order.begin(), inital_join_order.end(), 0);
	unique_join_orders.insert(inital_join_order);
	for (idx_t i = 0; i < SAMPLE_COUNT; i++) {
		if (unique_join_orders.size() == max_unique_join_orders) {
			break;
		}
		auto join_nodes = DpSize(node_infos, dependencies);
		if (join_nodes.size() != joins.size() + 1) {
			throw InternalException("SelSampleEnumeration: no complete join order (dependencies cannot be met)");
		}
		vector<idx_t> join_order(join_nodes.size() - 1);
		for (idx_t j = 1; j < join_nodes.size(); j++) {
			join_order[j - 1] = join_nodes[j] - 1; // node id = join index
		}
		unique_join_orders.insert(join_order);
		cost_map.clear();
		card_map.clear();
		best_plans.clear();
	}
	unique_join_orders.erase(inital_join_order);
	join_orders.reserve(unique_join_orders.size() + 1);
	join_orders.push_back(inital_join_order);
	join_orders.insert(join_orders.cend(), unique_join_orders.cbegin(), unique_join_orders.cend());
	nodes = nullptr;
}

} // namespace duckdb_polr
