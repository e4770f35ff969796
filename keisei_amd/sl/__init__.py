"""Supervised-learning path (mirror of keisei/sl/{dataset,trainer}.py): shard reader and SLTrainer."""
