"""Prompt templates of the navigation agents, byte-exact (pinned by tests/golden/g6_prompts.json):
tasks/agents/r2r.py:16-31, reverie.py:16-31,50-68, cvdn.py, soon.py.  Only the two task sentences differ
between agents."""

_NAV_TASK = {
    "r2r": "### Instruction: Navigate following the instruction. {} \n",
    "reverie": "### Instruction: Go to the location to complete the given task. Task: {} \n",
}
_NAV_HINT = {
    "r2r": "Compare the History and Instruction to infer your current progress, and then select the correct "
           "direction from the candidates to go to the target location.\n",
    "reverie": "Explore the scene to find out the targeted room and object. Then select the correct direction "
               "from the candidates to go to the target location.\n",
}


def navigation_prompt(agent, instruction, hist_num, cand_num, cls_token="<cls_1>"):
    p = _NAV_TASK[agent].format(instruction)
    p += "Following is the History, which contains the visual information of your previous decisions.\n"
    p += "### History: {}\n".format(" ".join("({}) <hist>".format(i) for i in range(hist_num)))
    p += ("Following is the Candidate, which contains several directions you can go to at the current position, "
          "candidate (0) is stop.\n")
    p += "### Candidate: {}\n".format(" ".join("({}) <cand>".format(i) if i > 0 else "(0) stop" for i in range(cand_num)))
    p += _NAV_HINT[agent]
    p += "### Output: {}".format(cls_token)
    return p
