// duckdb-polr_amd/host/polar_enumeration_algo.cpp -- the join-order enumerators of the host mirror.
//
// Behaviour (which orders, in which sequence; for SAMPLE: which random numbers are drawn when) follows
// src/parallel/polar_enumeration_algo.cpp of the reference -- every routine names the lines whose behaviour it
// reproduces.  The code is this repository's own: joins and plan nodes are numbered, sets of them are 64-bit masks,
// prerequisites are one mask per join, the walks run off explicit work lists, the DPsize table is flat.
#include "polar_enumeration_algo.hpp"

#include <algorithm>
#include <numeric>
#include <tuple>

namespace duckdb_polr {

static JoinMask Bit(idx_t i) {
	return (JoinMask)1 << i;
}

static JoinOrder IdentityOrder(idx_t n) {
	JoinOrder order(n);
	std::iota(order.begin(), order.end(), (idx_t)0);
	return order;
}

// ---- prerequisites ------------------------------------------------------------------------------------------------
Prerequisites::Prerequisites(idx_t n_joins_p, const DependencyMap &dependencies) : n_joins(n_joins_p), needs(n_joins_p, 0) {
	if (n_joins > 63) {
		throw InternalException("more than 63 multiplexed joins");
	}
	for (const auto &entry : dependencies) {
		if (entry.first < n_joins) {
			for (const idx_t required : entry.second) {
				needs[entry.first] |= Bit(required);
			}
		}
	}
}

vector<idx_t> Prerequisites::Candidates(JoinMask placed) const {
	vector<idx_t> next;
	for (idx_t j = 0; j < n_joins; j++) {
		if (!(placed & Bit(j)) && MayFollow(placed, j)) {
			next.push_back(j);
		}
	}
	return next;
}

// the two CanJoin forms of the reference's interface (polar_enumeration_algo.cpp:126-145), kept for callers of the mirror
bool JoinEnumerationAlgo::CanJoin(vector<idx_t> &r, idx_t s, DependencyMap &dependencies) {
	const auto found = dependencies.find(s);
	if (found == dependencies.end()) {
		return true;
	}
	return std::all_of(found->second.begin(), found->second.end(),
	                   [&](idx_t required) { return std::find(r.begin(), r.end(), required) != r.end(); });
}

bool JoinEnumerationAlgo::CanJoin(vector<idx_t> &r, vector<idx_t> &s, DependencyMap &dependencies) {
	return std::any_of(s.begin(), s.end(), [&](idx_t one) { return CanJoin(r, one, dependencies); });
}

// ---- candidate selection (behaviour: polar_enumeration_algo.cpp:13-77; ties go to the earlier candidate) -----------
idx_t RandomCandidateSelector::SelectNextCandidate(const JoinOrder &candidates, const JoinList &) {
	return candidates[rand() % candidates.size()]; // libc rand(), like the reference: not reproducible across libcs
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
	if (candidates.empty()) {
		return 0;
	}
	for (const idx_t j : candidates) {
		score_of.emplace(j, joins[j]->uncertainty_level * joins[j]->estimated_cardinality);
	}
	return *std::min_element(candidates.begin(), candidates.end(),
	                         [&](idx_t a, idx_t b) { return score_of.at(a) < score_of.at(b); });
}

// the next pick of `selector` out of `pool`, removed from it
static idx_t Draw(CandidateSelector &selector, JoinOrder &pool, const JoinList &joins) {
	const idx_t pick = selector.SelectNextCandidate(pool, joins);
	pool.erase(std::find(pool.begin(), pool.end(), pick));
	return pick;
}

// ---- which enumerator a session asks for (behaviour: polar_enumeration_algo.cpp:79-124) ----------------------------
unique_ptr<JoinEnumerationAlgo> JoinEnumerationAlgo::CreateEnumerationAlgo(ClientContext &context) {
	auto selector_for = [](JoinEnumerator e) -> unique_ptr<CandidateSelector> {
		switch (e) {
		case JoinEnumerator::DFS_RANDOM:
		case JoinEnumerator::BFS_RANDOM:
			return unique_ptr<CandidateSelector>(new RandomCandidateSelector());
		case JoinEnumerator::DFS_MIN_CARD:
		case JoinEnumerator::BFS_MIN_CARD:
			return unique_ptr<CandidateSelector>(new MinCardinalitySelector());
		default:
			return unique_ptr<CandidateSelector>(new UncertainCardinalitySelector());
		}
	};
	const JoinEnumerator wanted = context.config.join_enumerator;
	unique_ptr<JoinEnumerationAlgo> algo;
	switch (wanted) {
	case JoinEnumerator::SAMPLE:
		algo.reset(new SelSampleEnumeration());
		break;
	case JoinEnumerator::DFS_RANDOM:
	case JoinEnumerator::DFS_MIN_CARD:
	case JoinEnumerator::DFS_UNCERTAIN:
		algo.reset(new DFSEnumeration(selector_for(wanted)));
		break;
	case JoinEnumerator::BFS_RANDOM:
	case JoinEnumerator::BFS_MIN_CARD:
	case JoinEnumerator::BFS_UNCERTAIN:
		algo.reset(new BFSEnumeration(selector_for(wanted)));
		break;
	case JoinEnumerator::EACH_LAST_ONCE:
		algo.reset(new EachLastOnceEnumeration());
		break;
	case JoinEnumerator::EACH_FIRST_ONCE:
		algo.reset(new EachFirstOnceEnumeration());
		break;
	default:
		throw InternalException("unknown join enumerator");
	}
	algo->max_join_orders = context.config.max_join_orders;
	return algo;
}

// the base class: the original order only (behaviour: :147-150)
void JoinEnumerationAlgo::GenerateJoinOrders(const vector<idx_t> &hash_join_idxs, DependencyMap &, const JoinList &,
                                             vector<JoinOrder> &join_orders) {
	join_orders.reserve(max_join_orders);
	join_orders.push_back(IdentityOrder(hash_join_idxs.size()));
}

// The bank always starts with the optimizer's own order (behaviour shared by the two walks: :573-608, :717-747): if the
// walk found it, it moves to the front; if not, it is put there and the bank keeps at most max_join_orders entries.
static void OriginalOrderFirst(vector<JoinOrder> &bank, idx_t n_joins, idx_t max_join_orders) {
	const JoinOrder original = IdentityOrder(n_joins);
	const auto found = std::find(bank.begin(), bank.end(), original);
	if (found == bank.end()) {
		bank.insert(bank.begin(), original);
		if (bank.size() > max_join_orders) {
			bank.pop_back();
		}
	} else {
		std::rotate(bank.begin(), found, found + 1);
	}
}

// ---- depth first (behaviour: :152-189, :558-608) --------------------------------------------------------------------
// Every level of the reference's recursion is a frame here: the prefix it extends, the joins that may come next, still
// to be drawn one by one (a RANDOM selector draws between a child's subtree and its next sibling, so the draws stay
// lazy).  A level is entered only while the bank has room.
void DFSEnumeration::GenerateJoinOrders(const vector<idx_t> &hash_join_idxs, DependencyMap &dependencies, const JoinList &joins,
                                        vector<JoinOrder> &join_orders) {
	const idx_t n = hash_join_idxs.size();
	const Prerequisites pre(n, dependencies);
	struct Frame {
		JoinOrder prefix;
		JoinMask placed;
		JoinOrder undrawn;
	};
	join_orders.reserve(max_join_orders + 1);
	vector<Frame> stack;
	if (join_orders.size() < max_join_orders) {
		stack.push_back(Frame {JoinOrder(), 0, pre.Candidates(0)});
	}
	while (!stack.empty()) {
		if (stack.back().undrawn.empty()) {
			stack.pop_back();
			continue;
		}
		Frame &frame = stack.back();
		const idx_t pick = Draw(*selector, frame.undrawn, joins);
		JoinOrder longer = frame.prefix;
		longer.push_back(pick);
		const JoinMask placed = frame.placed | Bit(pick);
		if (longer.size() == n) {
			join_orders.push_back(std::move(longer));
		} else if (join_orders.size() < max_join_orders) {
			stack.push_back(Frame {std::move(longer), placed, pre.Candidates(placed)}); // (invalidates `frame`)
		}
	}
	OriginalOrderFirst(join_orders, n, max_join_orders);
}

// ---- breadth first (behaviour: :669-747) -----------------------------------------------------------------------------
// Partial orders wait in ONE ordered work list, keyed by (joins placed so far, rank among the siblings drawn with it,
// running number): shorter prefixes first, then the earlier-drawn sibling, then age.  A prefix fans out into at most
// 4, 3, 2, 1, 1, ... children by depth; a prefix one join short of complete takes its last join and joins the bank.
void BFSEnumeration::GenerateJoinOrders(const vector<idx_t> &hash_join_idxs, DependencyMap &dependencies, const JoinList &joins,
                                        vector<JoinOrder> &join_orders) {
	const idx_t n = hash_join_idxs.size();
	const Prerequisites pre(n, dependencies);
	struct Partial {
		JoinOrder placed_before;
		idx_t next;
	};
	using Key = std::tuple<idx_t, idx_t, idx_t>; // level, sibling rank, running number
	std::map<Key, Partial> work;
	idx_t running = 0;
	auto fan_out = [&](const JoinOrder &prefix, JoinOrder pool, idx_t width) {
		width = std::min<idx_t>(width, pool.size());
		for (idx_t rank = 0; rank < width; rank++) {
			const idx_t pick = Draw(*selector, pool, joins);
			work.emplace(Key(prefix.size(), rank, running++), Partial {prefix, pick});
		}
	};
	fan_out(JoinOrder(), pre.Candidates(0), 4);
	join_orders.reserve(max_join_orders + 1);
	while (join_orders.size() <= max_join_orders && !work.empty()) {
		Partial head = std::move(work.begin()->second);
		work.erase(work.begin());
		JoinOrder prefix = std::move(head.placed_before);
		prefix.push_back(head.next);
		JoinMask placed = 0;
		for (const idx_t j : prefix) {
			placed |= Bit(j);
		}
		const JoinOrder pool = pre.Candidates(placed);
		if (prefix.size() + 1 == n && pool.size() == 1) {
			prefix.push_back(pool.front());
			join_orders.push_back(std::move(prefix));
		} else {
			fan_out(prefix, pool, prefix.size() >= 3 ? 1 : 4 - prefix.size());
		}
	}
	OriginalOrderFirst(join_orders, n, max_join_orders);
}

// ---- each join last once / first once (behaviour: :610-667) ----------------------------------------------------------
// `order` joins the bank if every join in it comes after the joins it is keyed by
static void AddIfFeasible(const Prerequisites &pre, const JoinOrder &order, vector<JoinOrder> &bank) {
	JoinMask placed = 0;
	for (const idx_t j : order) {
		if (!pre.MayFollow(placed, j)) {
			return;
		}
		placed |= Bit(j);
	}
	bank.push_back(order);
}

// the original order, then every order that moves ONE join to the end (the last join stays: that is the original)
void EachLastOnceEnumeration::GenerateJoinOrders(const vector<idx_t> &hash_join_idxs, DependencyMap &dependencies,
                                                 const JoinList &joins, vector<JoinOrder> &join_orders) {
	JoinEnumerationAlgo::GenerateJoinOrders(hash_join_idxs, dependencies, joins, join_orders);
	const Prerequisites pre(hash_join_idxs.size(), dependencies);
	const JoinOrder original = join_orders.front();
	for (idx_t moved = 0; moved + 1 < original.size(); moved++) {
		JoinOrder order = original;
		std::rotate(order.begin() + moved, order.begin() + moved + 1, order.end());
		AddIfFeasible(pre, order, join_orders);
	}
}

// the original order, then every order that moves ONE join to the front
void EachFirstOnceEnumeration::GenerateJoinOrders(const vector<idx_t> &hash_join_idxs, DependencyMap &dependencies,
                                                  const JoinList &joins, vector<JoinOrder> &join_orders) {
	JoinEnumerationAlgo::GenerateJoinOrders(hash_join_idxs, dependencies, joins, join_orders);
	const Prerequisites pre(hash_join_idxs.size(), dependencies);
	const JoinOrder original = join_orders.front();
	for (idx_t moved = 1; moved < original.size(); moved++) {
		JoinOrder order = original;
		std::rotate(order.begin(), order.begin() + moved, order.begin() + moved + 1);
		AddIfFeasible(pre, order, join_orders);
	}
}

// ---- SAMPLE (behaviour: :248-556) --------------------------------------------------------------------------------------
static const double kSelectivitySteps[] = {0.0001, 0.001, 0.01, 0.1, 0.2, 0.4, 0.8};
static const idx_t kSelectivityStepCount = sizeof(kSelectivitySteps) / sizeof(kSelectivitySteps[0]);

JoinMask SelSampleEnumeration::MaskOf(const NodeSeq &seq) {
	JoinMask m = 0;
	for (const uint8_t id : seq) {
		m |= Bit(id);
	}
	return m;
}

// a plan node and, behind it, the nodes of its nested join order (CreateJoinOrderNodes recursing, :248-287)
uint8_t SelSampleEnumeration::AddNode(const JoinOrderNodeInfo &info) {
	if (nodes.size() >= 64) {
		throw InternalException("SelSampleEnumeration: more than 64 plan nodes");
	}
	const uint8_t id = (uint8_t)nodes.size();
	nodes.push_back(PlanNode {(double)info.base_table_card, info.predicate, info.unique, {}});
	return id;
}

// One sampled selectivity: a step of the ladder chosen by the sample itself plus a little of the sample (:412, :459).  The
// third place the reference samples (:464) casts before it multiplies -- `(idx_t) rand * size` -- and so always takes
// the ladder's first step; `index_from_sample == false` is that form.  Kept: it decides which orders SAMPLE finds.
double SelSampleEnumeration::SampleSelectivity(bool index_from_sample) {
	const double sample = dist(rng);
	const idx_t step = index_from_sample ? (idx_t)(sample * kSelectivityStepCount) : (idx_t)0;
	return kSelectivitySteps[step] + sample * kSelectivitySteps[0];
}

// Cost of a plan prefix = sum of the cardinalities of its intermediate results (:404-483); the cardinality of a node SET
// is fixed the first time any order reaches it (card_of), whichever order that was.
double SelSampleEnumeration::PlanCost(const NodeSeq &plan) {
	const auto known = cost_of.find(plan);
	if (known != cost_of.end()) {
		return known->second;
	}
	if (plan.size() == 1) {
		const PlanNode &node = nodes[plan.front()];
		double card = node.base_table_card;
		if (!node.nested.empty()) {
			// a build side that is itself a join tree: its cardinality is that of its own join order, costed prefix by
			// prefix (whose costs do not count towards this plan's: :401-409)
			NodeSeq inner;
			for (const uint8_t id : node.nested) {
				inner.push_back(id);
				PlanCost(inner);
				cost_of[inner] = 0;
			}
			card = card_of[MaskOf(inner)];
		} else if (node.predicate) {
			card *= SampleSelectivity(true);
		}
		card_of[MaskOf(plan)] = card;
		cost_of[plan] = 0;
		return 0;
	}
	const uint8_t last = plan.back();
	const NodeSeq before(plan.begin(), plan.end() - 1);
	const JoinMask before_set = MaskOf(before), last_set = Bit(last), whole = before_set | last_set;
	if (!card_of.count(before_set)) {
		PlanCost(before);
	}
	if (!card_of.count(last_set)) {
		PlanCost(NodeSeq {last});
	}
	double card = card_of[before_set];
	// the nodes of the new set that carry a predicate, and the plan's first node
	JoinMask filtered = Bit(plan.front());
	for (idx_t id = 0; id < nodes.size(); id++) {
		if ((whole & Bit(id)) && nodes[id].predicate) {
			filtered |= Bit(id);
		}
	}
	if (card_of.count(whole)) {
		card = card_of[whole];
	} else if (card_of.count(filtered)) {
		card = card_of[filtered];
	} else if (nodes[last].unique) {
		// joining a key: the result cannot fall below the largest cardinality already fixed for a superset
		double floor_card = 0;
		const int size = __builtin_popcountll(whole);
		for (const auto &entry : card_of) {
			if (__builtin_popcountll(entry.first) > size && (entry.first & whole) == whole) {
				floor_card = std::max(floor_card, entry.second);
			}
		}
		if (nodes[last].predicate) {
			card = floor_card + SampleSelectivity(true) * (card - floor_card);
		}
	} else {
		const double selectivity = SampleSelectivity(false);
		card *= card_of[last_set] * selectivity;
	}
	card_of[whole] = card;
	const double cost = cost_of[before] + card; // (an uncosted prefix counts 0, and is remembered as that)
	cost_of[plan] = cost;
	return cost;
}

// One round of DPsize (:323-390): best plan per join subset, subsets by size, every (subset, extra join) pair in the
// reference's order -- the s-subsets of the joins in lexicographic order of their sorted index lists, the extra join
// ascending -- because a coin (one draw) decides which of two competing plans is costed first, and costing draws samples.
JoinOrder SelSampleEnumeration::OneRound(const Prerequisites &pre) {
	const idx_t n = pre.n_joins;
	vector<NodeSeq> best((size_t)1 << n);
	vector<char> have((size_t)1 << n, 0);
	for (idx_t j = 0; j < n; j++) {
		if (pre.MayFollow(0, j)) {
			best[Bit(j)] = NodeSeq {0, (uint8_t)(1 + j)};
			have[Bit(j)] = 1;
		}
	}
	for (idx_t size = 1; size < n; size++) {
		// the size-subsets of {0..n-1} in lexicographic order: an index list advanced like an odometer
		vector<idx_t> members(size);
		std::iota(members.begin(), members.end(), (idx_t)0);
		while (true) {
			JoinMask subset = 0;
			bool startable = false; // some member may come first (CanJoin(empty, subset): any of them)
			for (const idx_t j : members) {
				subset |= Bit(j);
				startable = startable || pre.MayFollow(0, j);
			}
			for (idx_t extra = 0; extra < n; extra++) {
				if ((subset & Bit(extra)) || !startable || !pre.MayFollow(subset, extra) || !have[subset]) {
					continue;
				}
				NodeSeq longer = best[subset];
				longer.push_back((uint8_t)(1 + extra));
				const JoinMask grown = subset | Bit(extra);
				if (have[grown]) {
					const bool longer_first = std::round(dist(rng)) != 0;
					const double first = PlanCost(longer_first ? longer : best[grown]);
					const double second = PlanCost(longer_first ? best[grown] : longer);
					const double cost_longer = longer_first ? first : second, cost_kept = longer_first ? second : first;
					if (cost_longer < cost_kept) {
						best[grown] = longer;
					}
				} else {
					best[grown] = longer;
					have[grown] = 1;
				}
			}
			// next subset
			idx_t pos = size;
			while (pos > 0 && members[pos - 1] == n - size + (pos - 1)) {
				pos--;
			}
			if (pos == 0) {
				break;
			}
			members[pos - 1]++;
			for (idx_t i = pos; i < size; i++) {
				members[i] = members[i - 1] + 1;
			}
		}
	}
	const size_t all = ((size_t)1 << n) - 1;
	if (!have[all]) {
		throw InternalException("SelSampleEnumeration: no complete join order (dependencies cannot be met)");
	}
	JoinOrder order;
	for (size_t i = 1; i < best[all].size(); i++) {
		order.push_back((idx_t)best[all][i] - 1); // node 1 + j = join j
	}
	return order;
}

// max_join_orders rounds, each with fresh samples; the bank = the original order + the distinct winners in
// lexicographic order; stops early once every order that can differ has been seen (:492-556)
void SelSampleEnumeration::GenerateJoinOrders(const vector<idx_t> &hash_join_idxs, DependencyMap &dependencies, const JoinList &joins,
                                              vector<JoinOrder> &join_orders) {
	const idx_t n = joins.size();
	if (n > 16) {
		throw InternalException("SelSampleEnumeration: more than 16 joins");
	}
	const Prerequisites pre(n, dependencies);
	// plan nodes: the source, the build side of every join, then -- depth first -- the nested join orders
	nodes.clear();
	vector<const JoinOrderNodeInfo *> pending;
	AddNode(joins.front()->probe_source_info);
	pending.push_back(&joins.front()->probe_source_info);
	for (const PhysicalHashJoin *join : joins) {
		AddNode(join->build_side_info);
		pending.push_back(&join->build_side_info);
	}
	for (size_t at = 0; at < pending.size(); at++) { // (pending grows while nested orders are unfolded)
		const JoinOrderNodeInfo *info = pending[at];
		for (const JoinOrderNodeInfo &inner : info->nested_join_order) {
			nodes[at].nested.push_back(AddNode(inner));
			pending.push_back(&inner);
		}
	}
	idx_t movable = 0; // build sides whose position can matter: filtered, or not joined on a key
	for (idx_t j = 1; j <= n; j++) {
		movable += (nodes[j].predicate || !nodes[j].unique) ? 1 : 0;
	}
	idx_t distinct_orders_possible = 1;
	for (idx_t f = 2; f <= movable; f++) {
		distinct_orders_possible *= f;
	}
	const JoinOrder original = IdentityOrder(n);
	std::map<JoinOrder, bool> seen; // (ordered: the bank lists the winners lexicographically)
	seen[original] = true;
	for (idx_t round = 0; round < max_join_orders && seen.size() != distinct_orders_possible; round++) {
		seen[OneRound(pre)] = true;
		cost_of.clear();
		card_of.clear();
	}
	seen.erase(original);
	join_orders.reserve(seen.size() + 1);
	join_orders.push_back(original);
	for (const auto &entry : seen) {
		join_orders.push_back(entry.first);
	}
	nodes.clear();
}

} // namespace duckdb_polr
