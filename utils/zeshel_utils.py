"""ZeShEL data-format contract used by the eval entry points (restated from the reference's
utils/zeshel_utils.py:1-79): per-world entity / mention counts, the train/valid/test world lists and
the score-matrix file naming convention.  Pure data + path formatting, no compute."""

MAX_ENT_LENGTH = 128
MAX_MENT_LENGTH = 128
MAX_PAIR_LENGTH = 256

# world -> (split, number of entities, number of mentions)
_WORLDS = {
	"forgotten_realms": ("test", 15603, 1200), "lego": ("test", 10076, 1199), "star_trek": ("test", 34430, 4227),
	"yugioh": ("test", 10031, 3374),
	"american_football": ("train", 31929, 3898), "doctor_who": ("train", 40281, 8334), "fallout": ("train", 16992, 3286),
	"final_fantasy": ("train", 14044, 6041), "military": ("train", 104520, 13063), "pro_wrestling": ("train", 10133, 1392),
	"starwars": ("train", 87056, 11824), "world_of_warcraft": ("train", 27677, 1437),
	"coronation_street": ("valid", 17809, 1464), "elder_scrolls": ("valid", 21712, 4275), "ice_hockey": ("valid", 28684, 2233),
	"muppets": ("valid", 21344, 2028),
}
N_ENTS_ZESHEL = {w: v[1] for w, v in _WORLDS.items()}
N_MENTS_ZESHEL = {w: v[2] for w, v in _WORLDS.items()}


def get_zeshel_world_info():
	"""[(split, world), ...] in the reference's order: test, train, valid."""
	out = []
	for split in ("test", "train", "valid"):
		out += [(split, w) for w, v in _WORLDS.items() if v[0] == split]
	return out


def score_matrix_filename(res_dir, world, n_ment):
	"""{res_dir}/{world}/ment_to_ent_scores_n_m_{n_ment}_n_e_{N_ENTS[world]}_all_layers_False.pkl
	(the n_e in the NAME is the world constant even if the stored tensor has another shape)."""
	return f"{res_dir}/{world}/ment_to_ent_scores_n_m_{n_ment}_n_e_{N_ENTS_ZESHEL[world]}_all_layers_False.pkl"


def get_dataset_info(data_dir, res_dir, worlds, n_ment=100):
	datasets = {}
	for split, world in worlds:
		datasets[world] = {
			"ment_file": f"{data_dir}/processed/{split}_worlds/{world}_mentions.jsonl",
			"ent_file": f"{data_dir}/documents/{world}.json",
			"ent_tokens_file": f"{data_dir}/tokenized_entities/{world}_128_bert_base_uncased.npy",
		}
	if res_dir is not None:
		for world in N_ENTS_ZESHEL:
			nm = N_MENTS_ZESHEL[world] if n_ment is None else n_ment
			datasets[world]["crossenc_ment_to_ent_scores"] = score_matrix_filename(res_dir, world, nm)
			datasets[world]["crossenc_ment_and_ent_embeds"] = \
				f"{res_dir}/{world}/ment_and_ent_embeds_n_m_{nm}_n_e_{N_ENTS_ZESHEL[world]}_all_layers_False.pkl"
	return datasets
