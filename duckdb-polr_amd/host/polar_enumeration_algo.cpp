#include "polar_enumeration_algo.hpp"

#include <algorithm>
#include <limits>
#include <numeric>
#include <queue>

namespace duckdb_polr {

// polar_enumeration_algo.cpp:13-16 (libc rand(), like the reference: not reproducible across libcs)
idx_t RandomCandidateSelector::SelectNextCandidate(const std::vector<idx_t> &join_idxs,
                                                   const vector<PhysicalHashJoin *> &joins) {
	return join_idxs[rand() % join_idxs.size()];
}

// :18-30
idx_t MinCardinalitySelector::SelectNextCandidate(const std::vector<idx_t> &join_idxs,
                                                  const vector<PhysicalHashJoin *> &joins) {
	idx_t min_card = std::numeric_limits<idx_t>::max();
	idx_t selected_candidate = 0;
	for (idx_t i = 0; i < join_idxs.size(); i++) {
		auto *join = joins[join_idxs[i]];
		if (join->estimated_cardinality < min_card) {
			min_card = join->estimated_cardinality;
			selected_candidate = join_idxs[i];
		}
	}
	return selected_candidate;
}

// :57-77
idx_t UncertainCardinalitySelector::SelectNextCandidate(const std::vector<idx_t> &join_idxs,
                                                        const vector<PhysicalHashJoin *> &joins) {
	idx_t min_card = std::numeric_limits<idx_t>::max();
	idx_t selected_candidate = 0;
	for (auto join_idx : join_idxs) {
		auto *join = joins[join_idx];
		if (uncertainties.find(join_idx) == uncertainties.end()) {
			uncertainties[join_idx] = join->uncertainty_level * join->estimated_cardinality;
		}
		if (uncertainties[join_idx] < min_card) {
			min_card = uncertainties[join_idx];
			selected_candidate = join_idx;
		}
	}
	return selected_candidate;
}

// :79-124
unique_ptr<JoinEnumerationAlgo> JoinEnumerationAlgo::CreateEnumerationAlgo(ClientContext &context) {
	unique_ptr<JoinEnumerationAlgo> algo;
	switch (context.config.join_enumerator) {
	case JoinEnumerator::DFS_RANDOM:
		algo.reset(new DFSEnumeration(unique_ptr<CandidateSelector>(new RandomCandidateSelector())));
		break;
	case JoinEnumerator::DFS_MIN_CARD:
		algo.reset(new DFSEnumeration(unique_ptr<CandidateSelector>(new MinCardinalitySelector())));
		break;
	case JoinEnumerator::DFS_UNCERTAIN:
		algo.reset(new DFSEnumeration(unique_ptr<CandidateSelector>(new UncertainCardinalitySelector())));
		break;
	case JoinEnumerator::BFS_RANDOM:
		algo.reset(new BFSEnumeration(unique_ptr<CandidateSelector>(new RandomCandidateSelector())));
		break;
	case JoinEnumerator::BFS_MIN_CARD:
		algo.reset(new BFSEnumeration(unique_ptr<CandidateSelector>(new MinCardinalitySelector())));
		break;
	case JoinEnumerator::BFS_UNCERTAIN:
		algo.reset(new BFSEnumeration(unique_ptr<CandidateSelector>(new UncertainCardinalitySelector())));
		break;
	case JoinEnumerator::EACH_LAST_ONCE:
		algo.reset(new EachLastOnceEnumeration());
		break;
	case JoinEnumerator::EACH_FIRST_ONCE:
		algo.reset(new EachFirstOnceEnumeration());
		break;
	case JoinEnumerator::SAMPLE:
		// SelSampleEnumeration (:370-556) samples selectivities with libstdc++'s mt19937(1337) stream
		// over plan-tree statistics the host mirror does not carry; Pipeline::Ready's own fallback when
		// an enumerator yields < 2 orders is BFS_MIN_CARD (pipeline.cpp:216-225) -- used here directly.
		algo.reset(new BFSEnumeration(unique_ptr<CandidateSelector>(new MinCardinalitySelector())));
		break;
	default:
		throw InternalException("unknown join enumerator");
	}
	algo->max_join_orders = context.config.max_join_orders;
	return algo;
}

// :126-135
bool JoinEnumerationAlgo::CanJoin(vector<idx_t> &r, idx_t s, std::unordered_map<idx_t, vector<idx_t>> &dependencies) {
	auto &prereq = dependencies[s];
	for (const auto required_relation : prereq) {
		if (std::find(r.begin(), r.end(), required_relation) == r.end()) {
			return false;
		}
	}
	return true;
}

// :137-150: just the default join order
void JoinEnumerationAlgo::GenerateJoinOrders(const vector<idx_t> &hash_join_idxs,
                                             std::unordered_map<idx_t, vector<idx_t>> &dependencies,
                                             const vector<PhysicalHashJoin *> &joins,
                                             vector<vector<idx_t>> &join_orders) {
	std::vector<idx_t> default_path(hash_join_idxs.size());
	std::iota(default_path.begin(), default_path.end(), 0);
	join_orders.reserve(max_join_orders);
	join_orders.push_back(default_path);
}

// :152-189
void DFSEnumeration::GeneratePathsRecursive(const vector<PhysicalHashJoin *> &joins,
                                            std::unordered_map<idx_t, vector<idx_t>> &join_prerequisites,
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
static void MoveOriginalOrderFirst(vector<vector<idx_t>> &join_orders, idx_t k, idx_t max_join_orders) {
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
                                        std::unordered_map<idx_t, vector<idx_t>> &dependencies,
                                        const vector<PhysicalHashJoin *> &joins, vector<vector<idx_t>> &join_orders) {
	vector<idx_t> joins_left(hash_join_idxs.size());
	std::iota(joins_left.begin(), joins_left.end(), 0);
	join_orders.reserve(max_join_orders + 1);
	GeneratePathsRecursive(joins, dependencies, join_orders, vector<idx_t>(), joins_left);
	MoveOriginalOrderFirst(join_orders, hash_join_idxs.size(), max_join_orders);
}

// :610-638
void EachLastOnceEnumeration::GenerateJoinOrders(const vector<idx_t> &hash_join_idxs,
                                                 std::unordered_map<idx_t, vector<idx_t>> &dependencies,
                                                 const vector<PhysicalHashJoin *> &joins,
                                                 vector<vector<idx_t>> &join_orders) {
	JoinEnumerationAlgo::GenerateJoinOrders(hash_join_idxs, dependencies, joins, join_orders);
	auto default_path = join_orders.front();
	for (idx_t i = 0; i + 1 < default_path.size(); i++) {
		vector<idx_t> generated_path;
		for (idx_t j = 0; j < default_path.size(); j++) {
			if (j == i) {
				continue;
			}
			if (!CanJoin(generated_path, default_path[j], dependencies)) {
				break;
			}
			generated_path.push_back(default_path[j]);
		}
		if (generated_path.size() == default_path.size() - 1 && CanJoin(generated_path, default_path[i], dependencies)) {
			generated_path.push_back(default_path[i]);
			join_orders.push_back(generated_path);
		}
	}
}

// :640-667
void EachFirstOnceEnumeration::GenerateJoinOrders(const vector<idx_t> &hash_join_idxs,
                                                  std::unordered_map<idx_t, vector<idx_t>> &dependencies,
                                                  const vector<PhysicalHashJoin *> &joins,
                                                  vector<vector<idx_t>> &join_orders) {
	JoinEnumerationAlgo::GenerateJoinOrders(hash_join_idxs, dependencies, joins, join_orders);
	auto default_path = join_orders.front();
	for (idx_t i = 1; i < default_path.size(); i++) {
		vector<idx_t> generated_path;
		if (!CanJoin(generated_path, default_path[i], dependencies)) {
			continue;
		}
		generated_path.push_back(default_path[i]);
		for (idx_t j = 0; j < default_path.size(); j++) {
			if (j == i) {
				continue;
			}
			if (!CanJoin(generated_path, default_path[j], dependencies)) {
				break;
			}
			generated_path.push_back(default_path[j]);
		}
		if (generated_path.size() == default_path.size()) {
			join_orders.push_back(generated_path);
		}
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
                                                 std::unordered_map<idx_t, vector<idx_t>> &dependencies) {
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
                                        std::unordered_map<idx_t, vector<idx_t>> &dependencies,
                                        const vector<PhysicalHashJoin *> &joins, vector<vector<idx_t>> &join_orders) {
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

} // namespace duckdb_polr
